"""``LVISEval``: image-level federated mAP by visibility range / out-of-frame
on the GPU.  Same constructor, methods, public state and printed lines as the
reference class (tao_amodal/evaluation/lvis_amodal/eval.py:14-550); the
per-(image, category) Python loops are replaced by the HIP pipeline of
``tao_amodal_amd.engine`` over the non-empty cells.
"""
import logging
from collections import OrderedDict
from collections.abc import Sequence

import numpy as np

from ... import flatten
from .._core import (N_REC, N_THR, CellView, GpuRun, LazyIous, LazyPointers,
                     EvalConstants, restrict_to_params,
                     masked_mean, now, summaries, timed)
from .lvis import LVIS
from .results import LVISResults


class Params:
    def __init__(self, iou_type):
        """Params of the amodal LVIS evaluation (reference eval.py:553-583)."""
        self.img_ids = []
        self.cat_ids = []
        self.iou_thrs = np.linspace(
            0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
        self.rec_thrs = np.linspace(
            0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
        self.max_dets = 300
        self.visibility_rng = [[0, 1.0], [0, 0.1], [0.1, 0.8], [0.8, 1.0],
                               [0, 0.8], [0, 1.0]]  # last: out-of-frame
        self.visibility_rng_lbl = [
            "all", "highly-occluded", "partially-occluded", "highly-visible",
            "highly-and-partially-occluded", "out-of-frame"]
        self.use_cats = 1
        self.img_count_lbl = ["r", "c", "f"]
        self.iou_type = iou_type


class _EvalImgs(Sequence):
    """The reference's flat ``eval_imgs`` list, index = (cat, range, image),
    None for empty cells -- entries are rebuilt when read."""

    def __init__(self, view, n_img, n_rng, n_cat):
        self.view, self.n_img, self.n_rng, self.n_cat = view, n_img, n_rng, n_cat

    def __len__(self):
        return self.n_cat * self.n_rng * self.n_img

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        c, rem = divmod(i, self.n_rng * self.n_img)
        r, img = divmod(rem, self.n_img)
        k = self.view.cell_of(img, c)
        return None if k is None else self.view.entry(k, r)


class LVISEval:
    def __init__(self, lvis_gt, lvis_dt, iou_type="segm", device=None, dist=None):
        """lvis_gt: LVIS instance or annotation path; lvis_dt: LVISResults
        instance, result path or list of dicts; iou_type: only "bbox" is
        evaluated on this path (the reference's CLI passes "bbox").
        ``dist`` (evaluation/_dist.Ctx): this process is one rank of a
        multi-GPU evaluation and lvis_gt / lvis_dt are its share (images of one
        block of the sorted image ids); precision / recall come out complete
        and identical on every rank."""
        self.logger = logging.getLogger(__name__)
        if iou_type not in ["bbox", "segm"]:
            raise ValueError("iou_type: {} is not supported.".format(iou_type))
        if isinstance(lvis_gt, LVIS):
            self.lvis_gt = lvis_gt
        elif isinstance(lvis_gt, str):
            self.lvis_gt = LVIS(lvis_gt)
        else:
            raise TypeError("Unsupported type {} of lvis_gt.".format(lvis_gt))
        if isinstance(lvis_dt, LVISResults):
            self.lvis_dt = lvis_dt
        elif isinstance(lvis_dt, (str, list)):
            self.lvis_dt = LVISResults(self.lvis_gt, lvis_dt)
        else:
            raise TypeError("Unsupported type {} of lvis_dt.".format(lvis_dt))
        self.device = device if dist is None else dist.device
        self.dist = dist
        self.eval_imgs = []
        self.eval = {}
        self.params = Params(iou_type=iou_type)
        self.results = OrderedDict()
        self.ious = {}
        self.params.img_ids = sorted(self.lvis_gt.get_img_ids())
        self.params.cat_ids = sorted(self.lvis_gt.get_cat_ids())
        self._run = None
        self._cat_pos = None

    # ------------------------------------------------------------ stages
    def evaluate(self):
        self.logger.info("Running per image evaluation.")
        self.logger.info("Evaluate annotation type *{}*".format(self.params.iou_type))
        if self.params.iou_type not in ("bbox", "segm"):
            raise ValueError("Unknown iou_type for iou computation.")
        self.params.img_ids = list(np.unique(self.params.img_ids))
        constants = EvalConstants(self.params, Params(self.params.iou_type), "lvis")
        use_cats = bool(self.params.use_cats)
        with timed("flatten"):
            # params.img_ids / cat_ids subsets (reference eval.py:59-105)
            gt_cols, dt_cols, self._cat_pos = restrict_to_params(
                self.lvis_gt.columns, self.lvis_dt.columns_dt, "image",
                self.params.img_ids, self.params.cat_ids, use_cats)
            if gt_cols is not self.lvis_gt.columns and self.params.iou_type == "segm":
                raise NotImplementedError("iou_type='segm' with an image subset")
            # use_cats = 0: class-agnostic cells, one per image (reference
            # eval.py:125-128,147-166)
            # (built on the device: flatten_dev; flatten.py for the inputs it
            # does not cover)
            from ... import flatten_dev
            flat = flatten_dev.flatten_lvis(gt_cols, dt_cols,
                                            self.lvis_dt.max_dets,
                                            use_cats=use_cats, device=self.device,
                                            share=self.dist is not None and not self.dist.whole)
        if self.params.iou_type == "segm":
            with timed("masks"):
                flat.masks = self._masks(flat)
        self.flat = flat
        self.freq_groups = self._prepare_freq_group()
        if self.dist is not None:
            if self.params.iou_type != "bbox":
                raise NotImplementedError("multi-GPU runs evaluate iou_type='bbox'")
            from .._dist import DistRun
            self._run = DistRun(flat, self.dist, constants=constants, dt=dt_cols,
                                max_dets=self.lvis_dt.max_dets,
                                subset=gt_cols is not self.lvis_gt.columns)
        else:
            self._run = GpuRun(flat, self.device, constants=constants)
        self._run.evaluate()
        view = CellView(self._run, flat.img_ids, 0, "image_id", "visibility_rng",
                        self.params.visibility_rng)
        view.cat_pos = self._cat_pos
        cats = self.params.cat_ids if use_cats else [-1]
        self.ious = LazyIous(view, self.params.img_ids, cats)
        self.eval_imgs = _EvalImgs(view, len(self.params.img_ids),
                                   len(self.params.visibility_rng), len(cats))

    def _masks(self, flat):
        """_to_mask (reference eval.py:54-58, 70-73) for the annotations that
        are evaluated: every ground truth / detection of the cell tables goes
        through ann_to_rle (lvis.py:171-193) into a native mask batch; a
        detection without "segmentation" gets the polygon of its box
        (results.py:48-49)."""
        from ...masks import MaskBatch
        imgs = self.lvis_gt.imgs
        anns = self.lvis_gt.dataset["annotations"]
        out = {}

        def gt_items():
            for row in flat.gt_row.tolist():
                a = anns[row]
                im = imgs[a["image_id"]]
                yield a["segmentation"], im["height"], im["width"]

        def dt_items(raw):
            for row in flat.dt_row.tolist():
                r = raw[row]
                im = imgs[r["image_id"]]
                if "segmentation" in r:
                    seg = r["segmentation"]
                else:
                    x1, y1, w, h = r["bbox"]
                    x2, y2 = x1 + w, y1 + h
                    seg = [[x1, y1, x1, y2, x2, y2, x2, y1]]
                yield seg, im["height"], im["width"]

        for side, items in (("gt", gt_items()),
                            ("dt", dt_items(self.lvis_dt.raw_results))):
            batch = MaskBatch()
            batch.add_many(items)       # polygons: one native call, all cores
            out[side] = batch.arrays()
            batch.close()
        return out

    def _prepare_freq_group(self):
        groups = [[] for _ in self.params.img_count_lbl]
        from ...columns import FREQ_MISSING, FREQ_OTHER
        freq = np.asarray(self.flat.cat_freq)
        if self._cat_pos is not None:          # params.cat_ids, in the caller's order
            freq = freq[self._cat_pos]
        for idx, fr in enumerate(freq.tolist()):
            if fr == FREQ_MISSING:
                raise KeyError("frequency")       # reference: cat["frequency"]
            if fr == FREQ_OTHER:
                raise ValueError("category frequency is not in list")
            groups[self.params.img_count_lbl.index(chr(fr))].append(idx)
        return groups

    def accumulate(self):
        self.logger.info("Accumulating evaluation results.")
        if self._run is None:
            self.logger.warning("Please run evaluate first.")
            return
        self._run.accumulate()
        n_rng = len(self.params.visibility_rng)
        precision, recall = self._run.precision, self._run.recall
        if self._cat_pos is not None:
            # the category axis in the order of params.cat_ids
            precision = np.ascontiguousarray(precision[:, :, self._cat_pos])
            recall = np.ascontiguousarray(recall[:, self._cat_pos])
        self.eval = {
            "params": self.params,
            "counts": [len(self.params.iou_thrs), len(self.params.rec_thrs),
                       len(self.params.cat_ids) if self.params.use_cats else 1,
                       n_rng],
            "date": now(),
            "precision": precision,
            "recall": recall,
            "dt_pointers": LazyPointers(self._run, n_rng, (n_rng,), self._cat_pos),
        }

    def _summarize(self, summary_type, iou_thr=None, visibility_rng="all",
                   freq_group_idx=None):
        aidx = [i for i, lbl in enumerate(self.params.visibility_rng_lbl)
                if lbl == visibility_rng]
        if summary_type == "ap":
            s = self.eval["precision"]
            if iou_thr is not None:
                s = s[np.where(iou_thr == self.params.iou_thrs)[0]]
            if freq_group_idx is not None:
                s = s[:, :, self.freq_groups[freq_group_idx], aidx]
            else:
                s = s[:, :, :, aidx]
        else:
            s = self.eval["recall"]
            if iou_thr is not None:
                s = s[np.where(iou_thr == self.params.iou_thrs)[0]]
            s = s[:, :, aidx]
        return masked_mean(s)

    def summarize(self):
        if not self.eval:
            raise RuntimeError("Please run accumulate() first.")
        max_dets = self.params.max_dets
        S = self._summarize
        jobs = []
        for suffix, rng in (("", "all"), ("-HO", "highly-occluded"),
                            ("-PO", "partially-occluded"),
                            ("-HP", "highly-and-partially-occluded"),
                            ("-HV", "highly-visible"), ("-OOF", "out-of-frame")):
            jobs.append(("AP" + suffix, lambda rng=rng: S("ap", visibility_rng=rng)))
            jobs.append(("AP50" + suffix,
                         lambda rng=rng: S("ap", iou_thr=0.50, visibility_rng=rng)))
            jobs.append(("AP75" + suffix,
                         lambda rng=rng: S("ap", iou_thr=0.75, visibility_rng=rng)))
        for name, g in (("APr", 0), ("APc", 1), ("APf", 2)):
            jobs.append((name, lambda g=g: S("ap", freq_group_idx=g)))
        jobs.append(("AR@{}".format(max_dets), lambda: S("ar")))
        # the key keeps only the first letter of the label, so the three
        # "highly-*" ranges share "ARh@300" and the last one wins
        # (reference eval.py:497-499)
        for rng in ["highly-occluded", "partially-occluded", "highly-visible",
                    "highly-and-partially-occluded", "out-of-frame"]:
            jobs.append(("AR{}@{}".format(rng[0], max_dets),
                         lambda rng=rng: S("ar", visibility_rng=rng)))
        for key, value in summaries(jobs):
            self.results[key] = value

    def run(self):
        self.evaluate()
        self.accumulate()
        with timed("summarize"):
            self.summarize()

    def result_lines(self):
        template = (" {:<18} {} @[ IoU={:<9} | visibility={:>6s} | "
                    "maxDets={:>3d} catIds={:>3s}] = {:0.3f}")
        names = {"HO": "Highly Occluded (vis < 0.1)",
                 "PO": "Partially Occluded (0.1 < vis < 0.8)",
                 "HP": "Highly + Partially Occluded (vis < 0.8)",
                 "HV": "Highly Visible (vis > 0.8)"}
        lines = []
        for key, value in self.results.items():
            is_ap = "AP" in key
            if len(key) > 2 and key[2].isdigit():
                iou = "{:0.2f}".format(float(key[2:4]) / 100)
            else:
                iou = "{:0.2f}:{:0.2f}".format(self.params.iou_thrs[0],
                                               self.params.iou_thrs[-1])
            group = key[2] if len(key) > 2 and key[2] in ["r", "c", "f"] else "all"
            if len(key) > 2 and key[-2:] in names:
                vis = names[key[-2:]]
            elif len(key) > 2 and key[-3:] == "OOF":
                vis = "Out-of-Frame"
            else:
                vis = "all"
            lines.append(template.format(
                "Average Precision" if is_ap else "Average Recall",
                "(AP)" if is_ap else "(AR)", iou, vis, self.params.max_dets,
                group, value))
        return lines

    def print_results(self):
        for line in self.result_lines():
            print(line)

    def get_results(self):
        if not self.results:
            self.logger.warning("results is empty. Call run().")
        return self.results
