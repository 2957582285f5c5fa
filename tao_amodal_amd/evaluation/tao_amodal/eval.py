"""``TaoEval``: track-level federated mAP (3D IoU) on the GPU.  Same
constructor, methods, public state and logged lines as the reference class
(tao_amodal/evaluation/tao_amodal/eval.py:120-717); the per-(video, category)
Python loops are replaced by the HIP pipeline over the non-empty cells.
"""
import logging
from collections import OrderedDict
from collections.abc import Mapping

import numpy as np

from .._core import (N_REC, N_THR, CellView, GpuRun, LazyIous, LazyPointers,
                     EvalConstants,
                     masked_mean, now, summaries, timed)
from .results import TaoResults
from .tao import Tao


class Params:
    def __init__(self, iou_type, iou_3d_type="3d_iou"):
        """Params of the Tao evaluation (reference eval.py:720-757)."""
        self.vid_ids = []
        self.cat_ids = []
        self.iou_thrs = np.linspace(
            0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
        self.rec_thrs = np.linspace(
            0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01) + 1), endpoint=True)
        self.max_dets = 300
        self.area_rng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2],
                         [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2],
                         [0 ** 2, 1e5 ** 2]]
        self.area_rng_lbl = ["all", "small", "medium", "large",
                             "highly-and-partially-occluded"]
        self.time_rng = [[0, 1e5], [0, 3], [3, 10], [10, 1e5]]
        self.time_rng_lbl = ["all", "short", "medium", "long"]
        self.use_cats = 1
        self.vid_count_lbl = ["r", "c", "f"]
        self.iou_type = iou_type
        self.iou_3d_type = iou_3d_type


class _EvalVids(Mapping):
    """``eval_vids[(v, c, a, t)]`` of the reference, rebuilt when read."""

    def __init__(self, view, n_vid, n_cat, n_area, n_time):
        self.view = view
        self.shape = (n_vid, n_cat, n_area, n_time)

    def __len__(self):
        return int(np.prod(self.shape))

    def __iter__(self):
        V, C, A, T = self.shape
        return ((v, c, a, t) for c in range(C) for a in range(A)
                for t in range(T) for v in range(V))

    def __getitem__(self, key):
        v, c, a, t = key
        V, C, A, T = self.shape
        if not (0 <= v < V and 0 <= c < C and 0 <= a < A and 0 <= t < T):
            raise KeyError(key)
        k = self.view.cell_of(v, c)
        if k is None:
            return None
        e = self.view.entry(k, a * T + t)
        e["area_rng"], e["time_rng"] = e.pop("rng")
        return e


class TaoEval:
    def __init__(self, tao_gt, tao_dt, logger=None, iou_type="bbox",
                 iou_3d_type="3d_iou", device=None, dist=None):
        """``dist`` (evaluation/_dist.Ctx): this process is one rank of a
        multi-GPU evaluation, tao_gt / tao_dt are its share (videos of one
        block of the sorted video ids)."""
        if not logger:
            self.logger = logging.getLogger("tao.eval")
        elif isinstance(logger, str):
            self.logger = logging.getLogger(logger)
        else:
            self.logger = logger
        if iou_type not in ["bbox", "segm"]:
            raise ValueError("iou_type: {} is not supported.".format(iou_type))
        if isinstance(tao_gt, Tao):
            self.tao_gt = tao_gt
        elif isinstance(tao_gt, str):
            self.tao_gt = Tao(tao_gt)
        else:
            raise TypeError("Unsupported type {} of tao_gt.".format(tao_gt))
        if isinstance(tao_dt, TaoResults):
            self.tao_dt = tao_dt
        elif isinstance(tao_dt, (str, list)):
            self.tao_dt = TaoResults(self.tao_gt, tao_dt)
        else:
            raise TypeError("Unsupported type {} of tao_dt.".format(tao_dt))
        self.device = device if dist is None else dist.device
        self.dist = dist
        self.eval_vids = {}
        self.eval = {}
        self.params = Params(iou_type=iou_type, iou_3d_type=iou_3d_type)
        self.results = OrderedDict()
        self.ious = {}
        self.params.vid_ids = sorted(self.tao_gt.get_vid_ids())
        self.params.cat_ids = sorted(self.tao_gt.get_cat_ids())
        self.flat = self.tao_dt.flat
        self._run = None
        self._cat_pos = None

    # ------------------------------------------------------------ stages
    def evaluate(self, show_progress=False):
        self.logger.info("Running per video evaluation.")
        self.logger.info("Evaluate annotation type *{}*".format(self.params.iou_type))
        if self.params.iou_type != "bbox":
            raise NotImplementedError("only iou_type='bbox' runs on the HIP path")
        if self.params.iou_3d_type not in ("3d_iou", "avg_iou", "imagenetvid"):
            raise ValueError("Unknown iou_3d_type %r" % self.params.iou_3d_type)
        self.params.vid_ids = list(np.unique(self.params.vid_ids))
        constants = EvalConstants(self.params, Params(self.params.iou_type), "tao")
        # params.vid_ids / cat_ids subsets (reference eval.py:178-233)
        from .._core import restrict_to_params
        gt_cols, dt_cols, self._cat_pos = restrict_to_params(
            self.tao_gt.columns, self.tao_dt.columns_dt, "video",
            self.params.vid_ids, self.params.cat_ids, bool(self.params.use_cats))
        subset = gt_cols is not self.tao_gt.columns
        if subset:
            if len(gt_cols.ann_id) == 0:
                raise ValueError("Found no groundtruth annotations for given params")
            if len(dt_cols) == 0:
                raise ValueError("Found no predicted annotations for given params")
            if self.dist is not None:
                raise NotImplementedError("params.vid_ids subsets in a multi-GPU run")
        if not self.params.use_cats:
            # class-agnostic cells (reference eval.py:257-260,293-303)
            from ... import flatten
            self.flat = flatten.flatten_tao(
                gt_cols, dt_cols, self.tao_dt.max_dets, use_cats=False)
        elif subset:
            from ... import flatten_dev
            self.flat = flatten_dev.flatten_tao(gt_cols, dt_cols, self.tao_dt.max_dets,
                                                device=self.device)
        flat = self.flat
        if self.dist is not None:
            from .._dist import DistRun
            self._run = DistRun(flat, self.dist, self.params.iou_3d_type,
                                constants=constants)
        else:
            self._run = GpuRun(flat, self.device, self.params.iou_3d_type,
                               constants=constants)
        self._run.evaluate()
        P = self.params
        rngs = [(a, t) for a in P.area_rng for t in P.time_rng]
        view = CellView(self._run, flat.vid_ids, -1, "video_id", "rng", rngs)
        view.cat_pos = self._cat_pos
        cats = P.cat_ids if P.use_cats else [-1]
        self.ious = LazyIous(view, P.vid_ids, cats)
        self.eval_vids = _EvalVids(view, len(P.vid_ids), len(cats),
                                   len(P.area_rng), len(P.time_rng))

    @property
    def near_threshold_pairs(self):
        """Track pairs whose 3D IoU the frame-order guard recomputed in the
        reference's set order (it sat within the reordering bound of a
        comparison of the match).  Reads a device counter: synchronises."""
        if self._run is None:
            return 0
        return self._run.engine.guarded_pairs(self._run.dp, self._run.ws)

    def accumulate(self):
        self.logger.info("Accumulating evaluation results.")
        if self._run is None:
            self.logger.warning("Please run evaluate first.")
            return
        self._run.accumulate()
        # frame-order guard: the listed pairs are recomputed one per thread in
        # the reference's set order -- rare on real data (an IoU within the
        # reordering bound of a comparison); many of them (identical tracks,
        # avg_iou on integer boxes with exact rival ties) make the pass slow
        # without making it wrong: say so
        near, n_iou = self._run.near_threshold_pairs, getattr(self._run.dp, "n_iou", 0)
        if near > 4096 and near * 100 > n_iou:
            logging.getLogger("taoamd.guard").warning(
                "%d of %d track pairs went through the frame-order guard", near, n_iou)
        P = self.params
        A, T = len(P.area_rng), len(P.time_rng)
        K = len(P.cat_ids) if P.use_cats else 1
        precision, recall = self._run.precision, self._run.recall
        if self._cat_pos is not None:
            # the category axis in the order of params.cat_ids
            precision = np.ascontiguousarray(precision[:, :, self._cat_pos])
            recall = np.ascontiguousarray(recall[:, self._cat_pos])
        self.eval = {
            "params": P,
            "counts": [len(P.iou_thrs), len(P.rec_thrs), K, A, T],
            "date": now(),
            "precision": precision.reshape(len(P.iou_thrs), len(P.rec_thrs), K, A, T),
            "recall": recall.reshape(len(P.iou_thrs), K, A, T),
            "dt_pointers": LazyPointers(self._run, A * T, (A, T), self._cat_pos),
        }

    def _summarize(self, summary_type, iou_thr=None, area_rng="all",
                   time_rng="all", freq_group_idx=None):
        aidx = [i for i, lbl in enumerate(self.params.area_rng_lbl)
                if lbl == area_rng]
        tidx_ = [i for i, lbl in enumerate(self.params.time_rng_lbl)
                 if lbl == time_rng]
        if summary_type == "ap":
            s = self.eval["precision"]
            if iou_thr is not None:
                s = s[np.where(iou_thr == self.params.iou_thrs)[0]]
            s = s[:, :, :, aidx, tidx_]
        else:
            s = self.eval["recall"]
            if iou_thr is not None:
                s = s[np.where(iou_thr == self.params.iou_thrs)[0]]
            s = s[:, :, aidx, tidx_]
        return masked_mean(s)

    def summarize(self):
        if not self.eval:
            raise RuntimeError("Please run accumulate() first.")
        max_dets = self.params.max_dets
        S = self._summarize
        hp = "highly-and-partially-occluded"
        jobs = [("AP", lambda: S("ap")),
                ("AP50", lambda: S("ap", iou_thr=0.50)),
                ("AP75", lambda: S("ap", iou_thr=0.75)),
                ("AP-HP", lambda: S("ap", area_rng=hp)),
                ("AP50-HP", lambda: S("ap", area_rng=hp, iou_thr=0.50)),
                ("AP75-HP", lambda: S("ap", area_rng=hp, iou_thr=0.75))]
        for rng in ["small", "medium", "large"]:
            jobs.append((("AP", "area", rng, max_dets), lambda rng=rng: S("ap", area_rng=rng)))
        for rng in ["short", "medium", "long"]:
            jobs.append((("AP", "time", rng, max_dets), lambda rng=rng: S("ap", time_rng=rng)))
        jobs.append(("AR@{}".format(max_dets), lambda: S("ar")))
        for rng in ["small", "medium", "large"]:
            jobs.append((("AR", "area", rng, max_dets), lambda rng=rng: S("ar", area_rng=rng)))
        for rng in ["short", "medium", "long"]:
            jobs.append((("AR", "time", rng, max_dets), lambda rng=rng: S("ar", time_rng=rng)))
        for key, value in summaries(jobs):
            self.results[key] = value

    def run(self, show_progress=False):
        self.evaluate(show_progress=show_progress)
        self.accumulate()
        with timed("summarize"):
            self.summarize()

    def result_lines(self):
        template = (" {:<18} {} @[ IoU={:<9} | area={:>6s} | dur={:>6s} | "
                    "maxDets={:>3d} catIds={:>3s}] = {:0.3f}")
        lines = []
        for key, value in self.results.items():
            max_dets = self.params.max_dets
            is_ap = "AP" in key
            area = time = "all"
            if isinstance(key, tuple):
                kind, rng, max_dets = key[1:]
                if kind == "time":
                    time = rng[0]
                elif kind == "area":
                    area = rng[0]
                else:
                    raise ValueError("This should not happen")
            if len(key) > 2 and key[2].isdigit():
                iou = "{:0.2f}".format(float(key[2:4]) / 100)
            else:
                iou = "{:0.2f}:{:0.2f}".format(self.params.iou_thrs[0],
                                               self.params.iou_thrs[-1])
            group = key[2] if len(key) > 2 and key[2] in ["r", "c", "f"] else "all"
            lines.append(template.format(
                "Average Precision" if is_ap else "Average Recall",
                "(AP)" if is_ap else "(AR)", iou, area, time, max_dets, group,
                value))
        return lines

    def print_results(self):
        for line in self.result_lines():
            self.logger.info(line)

    def get_results(self):
        if not self.results:
            self.logger.warning("results is empty. Call run().")
        return self.results
