"""``LVISResults``: the prediction list seen by the image-level evaluator
(reference tao_amodal/evaluation/lvis_amodal/results.py:9-84).

The reference deep-copies the whole ground truth and rewrites every
prediction dict (area, id, polygon).  Here predictions are parsed once into
``DTColumns``; the top-300-per-image cut, ``area = w*h`` and
``id = position + 1`` are applied by ``flatten.flatten_lvis`` on arrays.
"""
import logging

import numpy as np

from ...columns import DTColumns
from ...flatten import limit_dets_per_image
from .lvis import LVIS


def _all_known(sorted_ids, values):
    """every value is one of sorted_ids (native: 30 M image ids on all
    threads; numpy when the reader library is not built)"""
    from ...columns import _ingest_lib
    try:
        lib = _ingest_lib()
    except OSError:
        from ...flatten import _lookup
        return bool((_lookup(sorted_ids, values) >= 0).all())
    import ctypes as C
    k = np.ascontiguousarray(sorted_ids, dtype=np.int64)
    v = np.ascontiguousarray(values, dtype=np.int64)
    lib.taoamd_host_all_in_sorted.restype = C.c_int
    lib.taoamd_host_all_in_sorted.argtypes = [C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    rc = lib.taoamd_host_all_in_sorted(len(k), k.ctypes.data, len(v), v.ctypes.data)
    assert rc >= 0
    return rc == 1


def merge_categories_like_reference(results, merge):
    """tao_amodal/results.py:47-50: merged categories mapped on EVERY dict of
    the caller's list, before anything is checked."""
    for x in results:
        if x["category_id"] in merge:
            x["category_id"] = merge[x["category_id"]]


def rewrite_like_reference(results, columns, max_dets, tao=False):
    """A LIST of prediction dicts handed to the constructor is the caller's:
    the reference rewrites it in place (results.py:39-65; tao_amodal/
    results.py:47-98) -- every box that survives the per-image cut gets
    ``segmentation`` (the box's polygon, where absent), ``area`` and ``id`` (1 +
    its place in the post-truncation list); the track level first maps merged
    categories on EVERY dict and, where a track's kept boxes carry different
    scores, replaces them by their np.mean -- and a caller may look at its
    dicts afterwards.  The evaluation itself runs on the columns; this loop is
    the reference's, on the dicts, for list inputs only.  ``tao``: the track
    level's form (its category map is merge_categories_like_reference)."""
    if not results:
        return
    keep = limit_dets_per_image(columns, max_dets) if max_dets >= 0 \
        else np.arange(len(results))
    kept = [results[k] for k in keep.tolist()]
    if "bbox" in kept[0]:
        for id_, ann in enumerate(kept):
            x1, y1, w, h = ann["bbox"]
            x2 = x1 + w
            y2 = y1 + h
            if "segmentation" not in ann:
                ann["segmentation"] = [[x1, y1, x1, y2, x2, y2, x2, y1]]
            ann["area"] = w * h
            ann["id"] = id_ + 1
    elif not tao and "segmentation" in kept[0]:
        # results given as compressed RLE (results.py:54-60): area and, where
        # absent, the tight box come from the mask
        area = np.asarray(columns.area)[keep]
        for id_, (k, ann) in enumerate(zip(keep.tolist(), kept)):
            ann["area"] = np.uint32(area[id_])
            if "bbox" not in ann:
                ann["bbox"] = np.array(columns.bbox[k], dtype=np.float64)
            ann["id"] = id_ + 1
    if tao:
        by_track = {}
        for ann in kept:
            by_track.setdefault(ann["track_id"], []).append(ann)
        for anns in by_track.values():
            scores = [float(x["score"]) for x in anns]
            if len(set(scores)) > 1:
                avg = np.mean(scores)
                for x in anns:
                    x["score"] = avg


class LVISResults(LVIS):
    def __init__(self, lvis_gt, results, max_dets=300, _share=False):
        """``_share``: the columns are one rank's share of a multi-GPU run (it
        may be empty; the whole list was checked in evaluation/_dist.py)."""
        if isinstance(lvis_gt, LVIS):
            self.gt = lvis_gt
        elif isinstance(lvis_gt, str):
            self.gt = LVIS(lvis_gt)
        else:
            raise TypeError("Unsupported type {} of lvis_gt.".format(lvis_gt))
        self.logger = logging.getLogger(__name__)
        self.logger.info("Loading and preparing results.")
        self._raw, self._raw_path = None, None
        if isinstance(results, DTColumns):
            self.columns_dt = results
        elif isinstance(results, str):
            self._raw_path = results
            try:
                self.columns_dt = DTColumns.from_json(results)
            except KeyError as e:
                if e.args != ("bbox",):
                    raise
                # no "bbox": results given as masks (reference results.py:54)
                self.columns_dt = self._from_masks(self.raw_results)
        else:
            self.logger.warning(
                "Assuming user provided the results in correct format.")
            assert isinstance(results, list), "results is not a list."
            self._raw = results
            if results and "bbox" not in results[0] \
                    and "segmentation" in results[0]:
                self.columns_dt = self._from_masks(results)
            else:
                self.columns_dt = DTColumns.from_json(results)
        self.max_dets = max_dets
        if len(self.columns_dt) == 0 and not _share:
            raise IndexError("list index out of range")  # results.py:42
        if self._raw is not None and self._raw_path is None:
            rewrite_like_reference(self._raw, self.columns_dt, max_dets)
        assert _all_known(np.unique(self.gt.columns.img_id), self.columns_dt.image_id), \
            "Results do not correspond to current LVIS set."
        self._index = None
        self._columns = None
        self._dataset = None

    @property
    def raw_results(self):
        """The prediction dicts (parsed on demand: only iou_type="segm" and
        mask-only results look at anything but the columns)."""
        if self._raw is None:
            if self._raw_path is None:
                raise ValueError("the prediction dicts are not available: "
                                 "LVISResults was built from columns")
            import json
            with open(self._raw_path, "r") as f:
                self._raw = json.load(f)
            assert isinstance(self._raw, list), "results is not a list."
        return self._raw

    @staticmethod
    def _from_masks(results):
        """Results that bring a compressed RLE instead of a box: area and,
        where absent, bbox come from the mask (reference results.py:54-60:
        mask_utils.area / toBbox)."""
        from ...masks import MaskBatch
        batch = MaskBatch()
        for r in results:
            seg = r["segmentation"]
            batch.add({"size": seg["size"], "counts": seg["counts"]}, 0, 0)
        arr = batch.arrays()
        batch.close()
        i64 = np.int64
        bbox = np.array([r["bbox"] if "bbox" in r else arr.bbox[k]
                         for k, r in enumerate(results)],
                        dtype=np.float64).reshape(-1, 4)
        cols = DTColumns(
            image_id=np.asarray([r["image_id"] for r in results], dtype=i64),
            category_id=np.asarray([r["category_id"] for r in results], dtype=i64),
            bbox=bbox,
            score=np.asarray([r["score"] for r in results], dtype=np.float64),
            track_id=np.asarray([r.get("track_id", -1) for r in results], dtype=i64),
            video_id=np.asarray([r.get("video_id", -1) for r in results], dtype=i64))
        cols.area = arr.area.astype(np.float64)
        return cols

    @property
    def dataset(self):
        """GT dataset with ``annotations`` replaced by the kept predictions
        (built on demand; the evaluation does not use it)."""
        if self._dataset is None:
            keep = limit_dets_per_image(self.columns_dt, self.max_dets)
            anns = self.columns_dt.take(keep).to_json()
            for k, a in enumerate(anns):
                x, y, w, h = a["bbox"]
                a["segmentation"] = [[x, y, x, y + h, x + w, y + h, x + w, y]]
                a["area"] = w * h
                a["id"] = k + 1
            self._dataset = dict(self.gt.dataset)
            self._dataset["annotations"] = anns
        return self._dataset

    def get_top_results(self, img_id, score_thrs):
        anns = self.load_anns(self.get_ann_ids(img_ids=[img_id]))
        return [a for a in anns if a["score"] > score_thrs]
