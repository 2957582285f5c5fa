"""Golden vectors of the reference evaluators run with EDITED evaluation
constants: ``params.iou_thrs`` / ``rec_thrs`` in other numbers and orders than
the defaults, and other VALUES in the range tables (``visibility_rng``;
``area_rng`` / ``time_rng``) -- public state the reference reads when it runs
(lvis_amodal/eval.py:143,205,234,319-322,407; tao_amodal/eval.py:272-275,385,
473-477,562).  Writes tests/golden/<name>/constants.npz: per case the
precision / recall tensors and the result values of both evaluators.

``CASES`` is a function of nothing: the test applies the same edits.
Development container only (needs /root/reference)."""
import json
import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402
from make_golden import reference_make_track_ids_unique  # noqa: E402


from constants_cases import cases, edit  # noqa: E402


def run(name):
    ref_lvis, ref_tao = refenv.import_reference()
    out = os.path.join(HERE, name)
    gt_path, pred_path = os.path.join(out, "gt.json"), os.path.join(out, "pred.json")
    lg = logging.getLogger("golden.constants")
    lg.propagate = False
    arrays = {}
    for cname, case in cases().items():
        def run_all(ev):
            """run(), but a summarize() that raises (fewer ranges than the
            labels name) is part of the golden behaviour."""
            ev.evaluate()
            ev.accumulate()
            try:
                ev.summarize()
                return ""
            except Exception as e:
                return type(e).__name__
        le = ref_lvis.LVISEval(gt_path, pred_path, "bbox")
        edit(le.params, case, "lvis")
        lerr = run_all(le)
        preds = json.load(open(pred_path))
        reference_make_track_ids_unique()(preds)
        te = ref_tao.TaoEval(ref_tao.Tao(gt_path), preds, logger=lg)
        edit(te.params, case, "tao")
        terr = run_all(te)
        arrays.update({
            cname + "_lvis_summarize_error": np.array(lerr),
            cname + "_tao_summarize_error": np.array(terr),
            cname + "_lvis_precision": le.eval["precision"],
            cname + "_lvis_recall": le.eval["recall"],
            cname + "_lvis_results": np.array([float(v) for v in le.results.values()]),
            cname + "_tao_precision": te.eval["precision"],
            cname + "_tao_recall": te.eval["recall"],
            cname + "_tao_results": np.array([float(v) for v in te.results.values()])})
        print(name, cname, "LVIS", le.eval["precision"].shape, le.results.get("AP"), lerr,
              "TAO", te.eval["precision"].shape, te.results.get("AP"), terr)
    np.savez_compressed(os.path.join(out, "constants.npz"), **arrays)


if __name__ == "__main__":
    for n in sys.argv[1:] or ["f1", "f4"]:
        run(n)
