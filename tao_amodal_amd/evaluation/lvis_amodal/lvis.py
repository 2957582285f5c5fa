"""``LVIS``: the annotation file of the image-level evaluator.

Mirror of the reference class (tao_amodal/evaluation/lvis_amodal/lvis.py:18-153)
for the part the bbox evaluation path uses.  The file is parsed once into
``GTColumns`` (struct of arrays); the dict indexes of the reference
(``anns``, ``imgs``, ``cats``, ``img_ann_map``, ``cat_img_map``) are built
lazily, only if somebody asks for them -- the evaluation itself never does.
"""
import json
import logging
from collections import defaultdict

from ...columns import GTColumns


class LVIS:
    def __init__(self, annotation_path, columns=None):
        """annotation_path: location of the annotation file (the reference
        accepts a path only; an already parsed dict is accepted as well).
        ``columns``: an already parsed GTColumns (a rank's share of the file in
        a multi-GPU run)."""
        self.logger = logging.getLogger(__name__)
        self.logger.info("Loading annotations.")
        self._columns = columns
        self._index = None
        self._dataset = None
        self._path = None
        if isinstance(annotation_path, dict):
            self._dataset = annotation_path
        else:
            # the native reader goes straight to columns; the dict form of the
            # file is only parsed if somebody asks for ``dataset``
            if self._columns is None and isinstance(annotation_path, str):
                self._columns = GTColumns.from_file_native(annotation_path)
            if self._columns is None:
                self._dataset = self._load_json(annotation_path)
            else:
                self._path = annotation_path
        assert self._columns is not None or type(self._dataset) == dict, (
            "Annotation file format {} not supported.".format(type(self._dataset)))

    def _load_json(self, path):
        with open(path, "r") as f:
            return json.load(f)

    @property
    def dataset(self):
        if self._dataset is None:
            self._dataset = self._load_json(self._path)
        return self._dataset

    @property
    def columns(self):
        if self._columns is None:
            self._columns = GTColumns.from_json(self.dataset)
        return self._columns

    # ------------------------------------------------- lazy dict indexes
    def _create_index(self):
        if self._index is None:
            self.logger.info("Creating index.")
            idx = {"anns": {}, "cats": {}, "imgs": {},
                   "img_ann_map": defaultdict(list),
                   "cat_img_map": defaultdict(list)}
            for ann in self.dataset["annotations"]:
                idx["img_ann_map"][ann["image_id"]].append(ann)
                idx["anns"][ann["id"]] = ann
                idx["cat_img_map"][ann["category_id"]].append(ann["image_id"])
            for img in self.dataset["images"]:
                idx["imgs"][img["id"]] = img
            for cat in self.dataset["categories"]:
                idx["cats"][cat["id"]] = cat
            self._index = idx
            self.logger.info("Index created.")
        return self._index

    anns = property(lambda self: self._create_index()["anns"])
    cats = property(lambda self: self._create_index()["cats"])
    imgs = property(lambda self: self._create_index()["imgs"])
    img_ann_map = property(lambda self: self._create_index()["img_ann_map"])
    cat_img_map = property(lambda self: self._create_index()["cat_img_map"])

    def get_ann_ids(self, img_ids=None, cat_ids=None, area_rng=None):
        """Same filter as the reference (lvis.py:63-97): strict area window."""
        anns = []
        if img_ids is not None:
            for img_id in img_ids:
                anns.extend(self.img_ann_map[img_id])
        else:
            anns = self.dataset["annotations"]
        if cat_ids is None and area_rng is None:
            return [a["id"] for a in anns]
        cat_ids = set(cat_ids)
        lo, hi = area_rng if area_rng is not None else (0, float("inf"))
        return [a["id"] for a in anns
                if a["category_id"] in cat_ids and lo < a["area"] < hi]

    def get_cat_ids(self):
        return self.columns.cat_id.tolist()

    def get_img_ids(self):
        return self.columns.img_id.tolist()

    def _load_helper(self, _dict, ids):
        return list(_dict.values()) if ids is None else [_dict[i] for i in ids]

    def load_anns(self, ids=None):
        return self._load_helper(self.anns, ids)

    def load_cats(self, ids):
        return self._load_helper(self.cats, ids)

    def load_imgs(self, ids):
        return self._load_helper(self.imgs, ids)

    def ann_to_rle(self, ann):
        """Polygons / uncompressed RLE of an annotation to a compressed RLE
        (reference lvis.py:171-193: mask_utils.frPyObjects + merge): a dict
        {"size": [h, w], "counts": bytes} as pycocotools returns it; an
        annotation that already holds a compressed RLE comes back as it is.
        Rasterisation, union and the text form are native (csrc/rle.cpp)."""
        img_data = self.imgs[ann["image_id"]]
        h, w = img_data["height"], img_data["width"]
        segm = ann["segmentation"]
        if not isinstance(segm, list) and not isinstance(segm["counts"], list):
            return segm
        from ...masks import MaskBatch
        batch = MaskBatch()
        try:
            batch.add(segm, h, w)
            text = batch.text(0)
        finally:
            batch.close()
        size = [int(h), int(w)] if isinstance(segm, list) else \
            [int(segm["size"][0]), int(segm["size"][1])]
        return {"size": size, "counts": text.encode("ascii")}

    def ann_to_mask(self, ann):
        """Binary mask of an annotation, uint8 [h, w] in Fortran order like
        mask_utils.decode (reference lvis.py:195-205)."""
        import numpy as np
        from ...masks import MaskBatch
        rle = self.ann_to_rle(ann)
        batch = MaskBatch()
        try:
            batch.add(rle, 0, 0)
            m = batch.arrays()
        finally:
            batch.close()
        h, w = int(m.hw[0, 0]), int(m.hw[0, 1])
        runs = m.counts[m.off[0]:m.off[1]].astype(np.int64)
        bits = np.repeat(np.arange(len(runs), dtype=np.uint8) & 1, runs)
        return np.asfortranarray(bits.reshape((h, w), order="F"))
