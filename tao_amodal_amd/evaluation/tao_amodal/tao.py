"""``Tao``: the annotation file of the track-level evaluator (reference
tao_amodal/evaluation/tao_amodal/tao.py:68-341).

Parsed once into ``GTColumns``.  The category merge (tao.py:98-118) is applied
inside ``flatten.flatten_tao`` on arrays; the reference's warnings about
merges and negative coordinates are emitted here, at construction, as it does.
"""
import json
import logging
from collections import defaultdict

import numpy as np

from ...columns import GTColumns
from ...flatten import count_bad_boxes


class Tao:
    def __init__(self, annotation_path, logger=None, columns=None):
        """``columns``: an already parsed GTColumns of the same file (lets the
        CLI share one parse between the two evaluators)."""
        if not logger:
            self.logger = logging.getLogger("tao.tao")
        elif isinstance(logger, str):
            self.logger = logging.getLogger(logger)
        else:
            self.logger = logger
        self.logger.info("Loading annotations.")
        self._columns = columns
        self._index = None
        self._dataset = None
        self._path = None
        if isinstance(annotation_path, dict):
            for key in ("info", "images", "annotations", "categories",
                        "videos", "tracks"):
                assert key in annotation_path, (
                    f"Provided dictionary does not contain key {key}")
            self._dataset = annotation_path
        else:
            # native reader -> columns; the dict form is parsed on demand
            if self._columns is None and isinstance(annotation_path, str):
                self._columns = GTColumns.from_file_native(annotation_path)
            if self._columns is None:
                self._dataset = self._load_json(annotation_path)
            else:
                self._path = annotation_path
        assert self._columns is not None or type(self._dataset) == dict, (
            "Annotation file format {} not supported.".format(type(self._dataset)))
        self._announce()

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

    @staticmethod
    def _construct_merge_map(dataset):
        merge_map = {}
        for category in dataset["categories"]:
            if "merged" in category:
                for to_merge in category["merged"]:
                    merge_map[to_merge["id"]] = category["id"]
        if not merge_map:
            logging.error("Did not merge any categories.")
        return merge_map

    def _announce(self):
        """The two diagnostics of the reference's _create_index
        (tao.py:104-105,143-158)."""
        self.logger.info("Creating index.")
        c = self.columns
        if len(c.cat_merged) == 0:
            logging.error("Did not merge any categories.")
        neg = count_bad_boxes(c.ann_bbox)
        if neg:
            self.logger.warning(f"{neg} annotations had negative values in "
                                f"coordinates!")
        self.logger.info("Index created.")

    # ------------------------------------------------- lazy dict indexes
    def _create_index(self):
        if self._index is None:
            ds = self.dataset
            idx = {"vids": {x["id"]: x for x in ds["videos"]},
                   "tracks": {x["id"]: x for x in ds["tracks"]},
                   "cats": {x["id"]: x for x in ds["categories"]},
                   "imgs": {x["id"]: x for x in ds["images"]},
                   "anns": {x["id"]: x for x in ds["annotations"]},
                   "vid_img_map": defaultdict(list),
                   "img_ann_map": defaultdict(list),
                   "track_ann_map": defaultdict(list)}
            for im in ds["images"]:
                idx["vid_img_map"][im["video_id"]].append(im)
            for a in ds["annotations"]:
                idx["img_ann_map"][a["image_id"]].append(a)
                idx["track_ann_map"][a["track_id"]].append(a)
            self._index = idx
        return self._index

    vids = property(lambda self: self._create_index()["vids"])
    tracks = property(lambda self: self._create_index()["tracks"])
    cats = property(lambda self: self._create_index()["cats"])
    imgs = property(lambda self: self._create_index()["imgs"])
    anns = property(lambda self: self._create_index()["anns"])
    vid_img_map = property(lambda self: self._create_index()["vid_img_map"])
    img_ann_map = property(lambda self: self._create_index()["img_ann_map"])
    track_ann_map = property(lambda self: self._create_index()["track_ann_map"])

    def get_cat_ids(self):
        return self.columns.cat_id.tolist()

    def get_vid_ids(self):
        return self.columns.vid_id.tolist()

    def get_img_ids(self):
        return self.columns.img_id.tolist()

    def _load_helper(self, _dict, ids):
        return list(_dict.values()) if ids is None else [_dict[i] for i in ids]

    def load_anns(self, ids=None):
        return self._load_helper(self.anns, ids)

    def load_tracks(self, ids=None):
        return self._load_helper(self.tracks, ids)

    def load_cats(self, ids):
        return self._load_helper(self.cats, ids)

    def load_imgs(self, ids):
        return self._load_helper(self.imgs, ids)

    def load_vids(self, ids):
        return self._load_helper(self.vids, ids)
