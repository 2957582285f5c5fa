"""Generate the golden vectors by running the REFERENCE evaluator.

Runs only in the development container (needs /root/reference, see
refenv.py).  For every fixture of fixtures.py it writes, under
tests/golden/<name>/ :

  gt.json, pred.json        the inputs (data, produced by fixtures.py)
  lvis.json.gz              per non-empty (image, category) cell of the
                            reference LVISEval: ious, and for each of the 6
                            visibility ranges dt_ids / gt_ids / dt_matches /
                            gt_matches / dt_ignore / gt_ignore / dt_scores;
                            dt_pointers (dt_ids, tps, fps); results; the 25
                            printed lines
  tao.json.gz               same for the reference TaoEval (20 area x time
                            ranges), plus the track ids after
                            make_track_ids_unique and the 19 printed lines
  eval.npz                  precision / recall of both evaluators restricted
                            to the categories that are not all -1
  cli_stdout.txt, cli_log.txt   output of the reference CLI on the pair

Usage:  python tests/golden/make_golden.py [fixture ...]
"""
import ast
import contextlib
import gzip
import io
import itertools
import json
import logging
import os
import subprocess
import sys
from collections import defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fixtures  # noqa: E402
import refenv  # noqa: E402

CLI = os.path.join(refenv.REF, "tools", "eval_on_tao_amodal.py")


def reference_make_track_ids_unique():
    """Compile just that one function out of the reference CLI (the module
    itself parses argv and runs the whole evaluation at import time)."""
    tree = ast.parse(open(CLI).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef)
          and n.name == "make_track_ids_unique"][0]
    ns = {"tqdm": lambda x: x, "itertools": itertools,
          "defaultdict": defaultdict}
    exec(compile(ast.Module([fn], []), CLI, "exec"), ns)
    return ns["make_track_ids_unique"]


def _l(a):
    return np.asarray(a).tolist()


def dump_cells(ious, evals, key_of, n_rng):
    """Group the reference's flat eval list by cell."""
    cells = {}
    for e in evals:
        if e is None:
            continue
        key = key_of(e)
        c = cells.setdefault(key[:2], {"key": list(key[:2]), "ranges": {}})
        c["ranges"][key[2]] = {
            "dt_ids": _l(e["dt_ids"]), "gt_ids": _l(e["gt_ids"]),
            "dt_scores": _l(e["dt_scores"]),
            "dt_matches": _l(e["dt_matches"]),
            "gt_matches": _l(e["gt_matches"]),
            "dt_ignore": _l(np.asarray(e["dt_ignore"]).astype(int)),
            "gt_ignore": _l(e["gt_ignore"]),
        }
    out = []
    for key, c in cells.items():
        assert len(c["ranges"]) == n_rng
        iou = ious[tuple(c["key"])]
        out.append({"key": [int(k) for k in c["key"]], "ious": _l(iou),
                    "ranges": [c["ranges"][r] for r in sorted(c["ranges"])]})
    return out


def pointers(dt_pointers, depth):
    out = []

    def rec(node, path):
        if len(path) == depth:
            if node:
                out.append({"idx": list(path), "dt_ids": _l(node["dt_ids"]),
                            "tps": _l(node["tps"].astype(int)),
                            "fps": _l(node["fps"].astype(int))})
            return
        for k in sorted(node):
            rec(node[k], path + [k])
    rec(dt_pointers, [])
    return out


def results_dict(res):
    return [[k if isinstance(k, str) else list(k), float(v)]
            for k, v in res.items()]


def run_fixture(name):
    ref_lvis, ref_tao = refenv.import_reference()
    out = os.path.join(HERE, name)
    os.makedirs(out, exist_ok=True)
    gt, pred = fixtures.ALL[name]()
    lite = name in fixtures.LITE
    # (big fixtures: the reference reads plain JSON from a scratch directory,
    # the committed copies are gzipped)
    work = os.path.join("/tmp", "golden_" + name) if lite else out
    os.makedirs(work, exist_ok=True)
    gt_path = os.path.join(work, "gt.json")
    pred_path = os.path.join(work, "pred.json")
    with open(gt_path, "w") as f:
        json.dump(gt, f, separators=(",", ":"))
    with open(pred_path, "w") as f:
        json.dump(pred, f, separators=(",", ":"))

    # ------------------------------------------------------------ LVISEval
    le = ref_lvis.LVISEval(gt_path, pred_path, "bbox")
    le.run()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        le.print_results()
    P = le.params
    n_img, n_rng = len(P.img_ids), len(P.visibility_rng)
    evals = []
    for c, cat in enumerate(P.cat_ids):
        for a in range(n_rng):
            for i in range(n_img):
                e = le.eval_imgs[(c * n_rng + a) * n_img + i]
                if e is not None:
                    e = dict(e)
                    e["_a"] = a
                evals.append(e)
    lvis = {
        "img_ids": [int(x) for x in P.img_ids],
        "cat_ids": [int(x) for x in P.cat_ids],
        "cells": None if lite else dump_cells(
            le.ious, evals, lambda e: (e["image_id"], e["category_id"],
                                       e["_a"]), n_rng),
        "dt_pointers": None if lite else pointers(le.eval["dt_pointers"], 2),
        # the integer match counts: [category index, range, detections,
        # TPs per threshold, FPs per threshold]
        "counts": [[p["idx"][0], p["idx"][1], len(p["dt_ids"]),
                    [int(np.sum(t)) for t in p["tps"]],
                    [int(np.sum(t)) for t in p["fps"]]]
                   for p in pointers(le.eval["dt_pointers"], 2)],
        "results": results_dict(le.results),
        "printed": buf.getvalue().splitlines(),
        "freq_groups": le.freq_groups,
    }
    lp, lr = le.eval["precision"], le.eval["recall"]

    # ------------------------------------------------------------- TaoEval
    preds = json.load(open(pred_path))
    n_changed = reference_make_track_ids_unique()(preds)
    uniq = [int(p["track_id"]) for p in preds]
    lines = []

    class H(logging.Handler):
        def emit(self, record):
            lines.append(record.getMessage())
    lg = logging.getLogger("golden." + name)
    lg.setLevel(logging.INFO)
    lg.propagate = False
    lg.handlers = [H()]
    te = ref_tao.TaoEval(ref_tao.Tao(gt_path), preds, logger=lg)
    te.run()
    lines.clear()
    te.print_results()
    TP = te.params
    n_vid = len(TP.vid_ids)
    evals = []
    for (v, c, a, t), e in te.eval_vids.items():
        if e is not None:
            e = dict(e)
            e["_a"] = a * len(TP.time_rng) + t
        evals.append(e)
    tao = {
        "vid_ids": [int(x) for x in TP.vid_ids],
        "cat_ids": [int(x) for x in TP.cat_ids],
        "n_track_ids_changed": int(n_changed),
        "unique_track_ids": uniq,
        "cells": dump_cells(te.ious, evals,
                            lambda e: (e["video_id"], e["category_id"],
                                       e["_a"]),
                            len(TP.area_rng) * len(TP.time_rng)),
        "dt_pointers": pointers(te.eval["dt_pointers"], 3),
        "results": results_dict(te.results),
        "printed": list(lines),
        "track_scores": {str(k): float(v["score"])
                         for k, v in te.tao_dt.tracks.items()},
    }
    tp, tr = te.eval["precision"], te.eval["recall"]

    def valid(p):
        return np.flatnonzero((p.reshape(p.shape[0], p.shape[1], p.shape[2],
                                         -1) > -1).any(axis=(0, 1, 3)))
    lk, tk = valid(lp), valid(tp)
    np.savez_compressed(
        os.path.join(out, "eval.npz"),
        lvis_valid_k=lk, lvis_precision=lp[:, :, lk], lvis_recall=lr[:, lk],
        lvis_shape=np.array(lp.shape),
        tao_valid_k=tk, tao_precision=tp[:, :, tk], tao_recall=tr[:, tk],
        tao_shape=np.array(tp.shape))
    for fn, obj in (("lvis.json.gz", lvis), ("tao.json.gz", tao)):
        with gzip.GzipFile(os.path.join(out, fn), "wb", mtime=0) as f:
            f.write(json.dumps(obj, separators=(",", ":")).encode())

    # ----------------------------------------------------------------- CLI
    log = os.path.join(out, "cli_log.txt")
    r = subprocess.run(
        [sys.executable, CLI, "--track_result", pred_path, "--annotation",
         gt_path, "--output_log", log],
        cwd=os.path.dirname(CLI), env=refenv.cli_env(),
        capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(os.path.join(out, "cli_stdout.txt"), "w") as f:
        f.write(r.stdout)
    # the two "Evaluating <path>" / "Loading gt <path>" lines carry the
    # absolute path of this checkout: make them location independent
    txt = open(log).read().replace(work + os.sep, "<DIR>/")
    with open(log, "w") as f:
        f.write(txt)
    if lite:
        for fn in ("gt.json", "pred.json"):
            with open(os.path.join(work, fn), "rb") as src, gzip.GzipFile(
                    os.path.join(out, fn + ".gz"), "wb", mtime=0) as dst:
                dst.write(src.read())
    sizes = {f: os.path.getsize(os.path.join(out, f))
             for f in sorted(os.listdir(out))}
    print(name, "LVIS AP", le.results["AP"], "TAO AP", te.results["AP"],
          sizes)


if __name__ == "__main__":
    names = sys.argv[1:] or list(fixtures.ALL)
    for n in names:
        run_fixture(n)
