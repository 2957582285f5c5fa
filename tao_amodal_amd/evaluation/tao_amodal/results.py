"""``TaoResults``: the prediction list seen by the track-level evaluator
(reference tao_amodal/evaluation/tao_amodal/results.py:11-132).

Predictions live in ``DTColumns``.  The checks the reference performs while
building its dicts are performed on arrays: one video per track id
(results.py:111-119), one category per track (:77-79), image ids known
(:105-109).  Per-image top-300, track scores (mean of the boxes' scores when
they differ, :88-98) and ``area = w*h`` happen in ``flatten.flatten_tao``.
"""
import logging

import numpy as np

from ... import flatten
from ...columns import DTColumns
from .tao import Tao


class TaoResults(Tao):
    def __init__(self, tao_gt, results, max_dets=300, _flat=None, _share=False):
        if isinstance(tao_gt, Tao):
            self.gt = tao_gt
        elif isinstance(tao_gt, str):
            self.gt = Tao(tao_gt)
        else:
            raise TypeError("Unsupported type {} of tao_gt.".format(type(tao_gt)))
        self.logger = logging.getLogger("tao.results")
        self.logger.info("Loading and preparing results.")
        raw = None
        if isinstance(results, DTColumns):
            self.columns_dt = results
        elif isinstance(results, str):
            self.columns_dt = DTColumns.from_json(results)
        else:
            self.logger.warning(
                "Assuming user provided the results in correct format.")
            assert isinstance(results, list), "results is not a list."
            self.columns_dt = DTColumns.from_json(results)
            raw = results
            # (the caller's dicts: results.py:47-50, before any check)
            from ..lvis_amodal.results import merge_categories_like_reference
            merge_categories_like_reference(
                raw, {int(a): int(b) for a, b in self.gt.columns.cat_merged.tolist()})
        self.max_dets = max_dets
        if len(self.columns_dt) == 0 and not _share:
            raise IndexError("list index out of range")  # results.py:61
        if len(self.gt.columns.cat_merged) == 0:
            # TaoResults rebuilds the merge map twice (results.py:47 and, via
            # _create_index, tao.py:115): two more root-logger records
            logging.error("Did not merge any categories.")
            logging.error("Did not merge any categories.")
        # cell tables of the track-level problem; this is where the track
        # scores are formed, as in the reference constructor
        from .._core import timed
        # (_flat: the same tables prepared ahead of time by the CLI, which
        # builds them on a worker thread while the image-level pass runs)
        with timed("flatten"):
            self.flat = _flat if _flat is not None else self._flatten(max_dets)
        neg = self.flat.get("neg_coords")      # counted by the device build
        if neg is None:
            b = self.columns_dt.bbox
            bad = (b[:, 0] < 0) | (b[:, 1] < 0) | (b[:, 2] <= 0) | (b[:, 3] <= 0)
            neg = 0
            if bad.any():
                keep = flatten.limit_dets_per_image(self.columns_dt, max_dets)
                neg = int(np.count_nonzero(bad[keep]))
        if neg:
            self.logger.warning(
                f"{neg} annotations had negative values in coordinates!")
        if self.flat.required_average:
            self.logger.warning(
                "At least one track had annotations with different scores; "
                "using average of individual annotation scores as track "
                "scores.")
        if raw is not None:
            # the caller's dicts, rewritten as the reference rewrites them
            # (results.py:47-98) -- after the checks above, which raise at the
            # reference's places before it would have touched the kept boxes
            from ..lvis_amodal.results import rewrite_like_reference
            rewrite_like_reference(raw, self.columns_dt, max_dets, tao=True)

    def _flatten(self, max_dets):
        """Cell tables on the device; inputs the reference rejects go through
        the numpy statement, which raises at the reference's places (one video
        per track first, results.py:111-119)."""
        from ... import flatten_dev
        dev = flatten_dev._cuda(None)
        if dev is not None:
            try:
                return flatten_dev.flatten_tao_device(
                    self.gt.columns, self.columns_dt, dev, max_dets)
            except (flatten_dev.Unsupported, flatten_dev.Rejected):
                pass
        self.ensure_unique_track_ids(self.columns_dt)
        return flatten.flatten_tao(self.gt.columns, self.columns_dt, max_dets)

    @staticmethod
    def ensure_unique_track_ids(dt):
        if len(dt) == 0:
            return
        u, first, inv = flatten.first_inverse(dt.track_id)
        bad = np.flatnonzero(dt.video_id != dt.video_id[first][inv])
        if len(bad):
            t = int(dt.track_id[bad[0]])
            raise AssertionError(
                f"Track id {t} appears in more than one video: "
                f"{int(dt.video_id[first][inv][bad[0]])} and "
                f"{int(dt.video_id[bad[0]])}")

    def get_top_results(self, img_id, score_thrs):
        raise NotImplementedError(
            "Unclear if this should be per image or per video")
