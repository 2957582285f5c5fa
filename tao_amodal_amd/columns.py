"""Columnar (struct-of-arrays) form of the two inputs of the evaluation path.

The reference keeps both inputs as lists of Python dicts and indexes them with
dict-of-list maps (reference: tao_amodal/evaluation/tao_amodal/tao.py:112-160,
tao_amodal/evaluation/lvis_amodal/lvis.py:38-61).  Here each JSON table becomes
a handful of contiguous numpy arrays so that every later stage (flatten, upload
to HBM, kernels) is index arithmetic over arrays instead of dict lookups.

Input contract (unchanged from the reference, README.md:107-116 and
tao.py:4-60):

* ground truth: ``{info, images, videos, tracks, annotations, categories}``
* predictions : list of ``{image_id, category_id, bbox, score, track_id,
  video_id}``

Ragged ``neg_category_ids`` / ``not_exhaustive_category_ids`` lists are kept
as CSR (offsets + values).
"""
import json
import os

import numpy as np


FREQ_MISSING, FREQ_OTHER = ord("?"), 0xFF

# A fresh process: {"event": threading.Event set when the HIP runtime has been
# started ahead of torch's import, "ok": it succeeded} -- set by the drop-in CLI
# (tools/eval_on_tao_amodal.py, _early_hip_init); DTColumns._from_file_device
EARLY_HIP = {}


def _ingest_lib():
    import ctypes as C
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                      "libtao_amodal_ingest.so")
    lib = C.CDLL(so)
    lib.taoamd_pred_write.argtypes = [C.c_char_p, C.c_int64] + [C.c_void_p] * 6
    lib.taoamd_gt_write.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int32]
    return lib


def _freq_byte(cat):
    if "frequency" not in cat:
        return FREQ_MISSING
    f = cat["frequency"]
    return ord(f) if isinstance(f, str) and len(f) == 1 and ord(f) < 0x80 \
        and f != "?" else FREQ_OTHER


def _csr(lists):
    off = np.zeros(len(lists) + 1, dtype=np.int64)
    if len(lists):
        off[1:] = np.cumsum([len(x) for x in lists])
    flat = [c for x in lists for c in x]
    return off, np.asarray(flat, dtype=np.int64).reshape(-1)


def _uncsr(off, val):
    return [val[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]


class GTColumns:
    """Ground-truth annotation file as arrays."""

    FIELDS = (
        "cat_id", "cat_freq", "cat_merged",
        "vid_id", "vid_neg_off", "vid_neg", "vid_nel_off", "vid_nel",
        "img_id", "img_vid", "img_frame",
        "img_neg_off", "img_neg", "img_nel_off", "img_nel",
        "trk_id", "trk_cat", "trk_vid", "trk_ignore",
        "ann_id", "ann_img", "ann_trk", "ann_cat", "ann_bbox", "ann_area",
        "ann_vis", "ann_oof", "ann_ignore",
    )

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw[f])

    # ------------------------------------------------------------------ json
    @classmethod
    def from_file_native(cls, path):
        """Parse an annotation file with the native columnar reader
        (csrc/ingest.cpp: one walk of the document, the five tables read in
        parallel); returns None when that library is not built.  Errors are
        those ``json.load`` + ``from_json`` would raise."""
        import ctypes as C
        import os
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                          "libtao_amodal_ingest.so")
        if not os.path.exists(so):
            return None
        lib = C.CDLL(so)
        if not hasattr(lib, "taoamd_gt_parse"):
            return None
        lib.taoamd_gt_parse.restype = C.c_void_p
        lib.taoamd_gt_parse.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        lib.taoamd_gt_array.argtypes = [C.c_void_p, C.c_char_p,
                                        C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        lib.taoamd_gt_free.argtypes = [C.c_void_p]
        lib.taoamd_gt_copy.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        err = C.create_string_buffer(512)
        h = lib.taoamd_gt_parse(os.fsencode(path), err, 512)
        if not h:
            msg = err.value.decode()
            if msg.startswith("not a dict"):
                kind = list if msg.endswith("list") else object
                raise AssertionError(
                    "Annotation file format {} not supported.".format(kind))
            if msg.startswith("cannot open"):
                raise FileNotFoundError(msg)
            if msg.startswith("KeyError: "):
                raise KeyError(msg[len("KeyError: "):].strip("'"))
            raise ValueError("malformed annotation file: " + msg)
        try:
            kw = {}
            for f in cls.FIELDS:
                ptr, n, el = C.c_void_p(), C.c_int64(), C.c_int()
                if lib.taoamd_gt_array(h, f.encode(), C.byref(ptr), C.byref(n),
                                       C.byref(el)):
                    raise RuntimeError("native reader has no array " + f)
                dtype = {8: np.int64, -8: np.float64, 1: np.uint8}[el.value]
                kw[f] = np.empty(n.value, dtype=dtype)
                if n.value:
                    lib.taoamd_gt_copy(h, f.encode(), kw[f].ctypes.data)
            kw["cat_merged"] = kw["cat_merged"].reshape(-1, 2)
            kw["ann_bbox"] = kw["ann_bbox"].reshape(-1, 4)
        finally:
            lib.taoamd_gt_free(h)
        return cls(**kw)

    @classmethod
    def from_json(cls, dataset):
        """dataset: parsed annotation dict (or a path)."""
        if isinstance(dataset, str):
            native = cls.from_file_native(dataset)
            if native is not None:
                return native
            with open(dataset, "r") as f:
                dataset = json.load(f)
        assert type(dataset) == dict, (
            "Annotation file format {} not supported.".format(type(dataset)))
        cats = dataset["categories"]
        vids = dataset["videos"]
        imgs = dataset["images"]
        trks = dataset["tracks"]
        anns = dataset["annotations"]
        merged = [(m["id"], c["id"]) for c in cats if "merged" in c
                  for m in c["merged"]]
        vneg = _csr([v["neg_category_ids"] for v in vids])
        vnel = _csr([v["not_exhaustive_category_ids"] for v in vids])
        ineg = _csr([i["neg_category_ids"] for i in imgs])
        inel = _csr([i["not_exhaustive_category_ids"] for i in imgs])
        i64 = np.int64
        return cls(
            cat_id=np.asarray([c["id"] for c in cats], dtype=i64),
            # one byte per category: the letter; FREQ_MISSING when the key is
            # absent, FREQ_OTHER for any other text ("rare", "") -- the
            # reference's img_count_lbl.index() fails on those
            cat_freq=np.asarray([_freq_byte(c)
                                 for c in cats], dtype=np.uint8),
            cat_merged=np.asarray(merged, dtype=i64).reshape(-1, 2),
            vid_id=np.asarray([v["id"] for v in vids], dtype=i64),
            vid_neg_off=vneg[0], vid_neg=vneg[1],
            vid_nel_off=vnel[0], vid_nel=vnel[1],
            img_id=np.asarray([i["id"] for i in imgs], dtype=i64),
            img_vid=np.asarray([i["video_id"] for i in imgs], dtype=i64),
            img_frame=np.asarray([i["frame_index"] for i in imgs],
                                 dtype=np.float64),
            img_neg_off=ineg[0], img_neg=ineg[1],
            img_nel_off=inel[0], img_nel=inel[1],
            trk_id=np.asarray([t["id"] for t in trks], dtype=i64),
            trk_cat=np.asarray([t["category_id"] for t in trks], dtype=i64),
            trk_vid=np.asarray([t["video_id"] for t in trks], dtype=i64),
            trk_ignore=np.asarray([1 if t.get("ignore", 0) else 0
                                   for t in trks], dtype=np.uint8),
            ann_id=np.asarray([a["id"] for a in anns], dtype=i64),
            ann_img=np.asarray([a["image_id"] for a in anns], dtype=i64),
            ann_trk=np.asarray([a["track_id"] for a in anns], dtype=i64),
            ann_cat=np.asarray([a["category_id"] for a in anns], dtype=i64),
            ann_bbox=np.asarray([a["bbox"] for a in anns],
                                dtype=np.float64).reshape(-1, 4),
            ann_area=np.asarray([a["area"] for a in anns], dtype=np.float64),
            ann_vis=np.asarray([a["visibility"] for a in anns],
                               dtype=np.float64),
            ann_oof=np.asarray([1 if a["out_of_frame"] else 0 for a in anns],
                               dtype=np.uint8),
            ann_ignore=np.asarray([1 if a.get("ignore", 0) else 0
                                   for a in anns], dtype=np.uint8),
        )

    @classmethod
    def concat(cls, parts):
        """Union of several annotation sets with disjoint ids over the same
        category table (used to assemble a multi-shard synthetic dataset)."""
        def cat(name):
            return np.concatenate([getattr(p, name) for p in parts])

        def csr(off, val):
            offs, base = [np.zeros(1, dtype=np.int64)], 0
            for p in parts:
                o = getattr(p, off)
                offs.append(o[1:] + base)
                base += int(o[-1])
            return np.concatenate(offs), cat(val)
        first = parts[0]
        vneg, vnel = csr("vid_neg_off", "vid_neg"), csr("vid_nel_off", "vid_nel")
        ineg, inel = csr("img_neg_off", "img_neg"), csr("img_nel_off", "img_nel")
        kw = dict(cat_id=first.cat_id, cat_freq=first.cat_freq,
                  cat_merged=first.cat_merged,
                  vid_neg_off=vneg[0], vid_neg=vneg[1],
                  vid_nel_off=vnel[0], vid_nel=vnel[1],
                  img_neg_off=ineg[0], img_neg=ineg[1],
                  img_nel_off=inel[0], img_nel=inel[1])
        for f in cls.FIELDS:
            if f not in kw:
                kw[f] = cat(f)
        return cls(**kw)

    def write_json(self, path):
        """The annotation file written by the native writer (csrc/
        jsonwrite.cpp): what ``json.dump(self.to_json())`` holds, without the
        dicts -- for sets of millions of annotations."""
        import ctypes as C
        lib = _ingest_lib()
        arrs = [np.ascontiguousarray(getattr(self, f)) for f in self.FIELDS]
        want = {"cat_freq": np.uint8, "trk_ignore": np.uint8, "ann_oof": np.uint8,
                "ann_ignore": np.uint8, "img_frame": np.float64,
                "ann_bbox": np.float64, "ann_area": np.float64,
                "ann_vis": np.float64}
        arrs = [np.ascontiguousarray(a, dtype=want.get(f, np.int64))
                for f, a in zip(self.FIELDS, arrs)]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        counts = (C.c_int64 * len(arrs))(*[a.size for a in arrs])
        rc = lib.taoamd_gt_write(os.fsencode(path), ptrs, counts, len(arrs))
        if rc:
            raise OSError("taoamd_gt_write(%s) failed: %d" % (path, rc))

    def select_videos(self, keep):
        """The annotation set of the videos `keep` (boolean mask over
        ``vid_id``): their images, tracks and annotations, the category table
        whole.  (A rank's share of one set split by video.)"""
        keep = np.asarray(keep, dtype=bool)
        vids = np.sort(self.vid_id[keep])

        def of(col):
            if len(vids) == 0:
                return np.zeros(len(col), bool)
            pos = np.searchsorted(vids, col)
            pos[pos == len(vids)] = 0
            return vids[pos] == col
        return self._select(keep, of(self.img_vid), of(self.trk_vid))

    def select_images(self, keep):
        """The annotation set of the images `keep` (boolean mask over
        ``img_id``) with their annotations; videos, tracks and categories
        whole.  (A rank's share of the image level, split by image id.)"""
        return self._select(np.ones(len(self.vid_id), bool), np.asarray(keep, bool),
                            np.ones(len(self.trk_id), bool))

    def _select(self, mv, mi, mt):
        def csr(off, val, m):
            lens = np.diff(off)[m]
            new_off = np.zeros(len(lens) + 1, dtype=np.int64)
            np.cumsum(lens, out=new_off[1:])
            idx = np.repeat(off[:-1][m] - new_off[:-1], lens) + np.arange(new_off[-1])
            return new_off, val[idx]
        imgs = np.sort(self.img_id[mi])
        if len(imgs):
            pos = np.searchsorted(imgs, self.ann_img)
            pos[pos == len(imgs)] = 0
            ma = imgs[pos] == self.ann_img
        else:
            ma = np.zeros(len(self.ann_img), bool)
        kw = dict(cat_id=self.cat_id, cat_freq=self.cat_freq, cat_merged=self.cat_merged)
        kw["vid_id"] = self.vid_id[mv]
        kw["vid_neg_off"], kw["vid_neg"] = csr(self.vid_neg_off, self.vid_neg, mv)
        kw["vid_nel_off"], kw["vid_nel"] = csr(self.vid_nel_off, self.vid_nel, mv)
        kw["img_neg_off"], kw["img_neg"] = csr(self.img_neg_off, self.img_neg, mi)
        kw["img_nel_off"], kw["img_nel"] = csr(self.img_nel_off, self.img_nel, mi)
        for f in self.FIELDS:
            if f in kw:
                continue
            m = mi if f.startswith("img_") else mt if f.startswith("trk_") else ma
            kw[f] = getattr(self, f)[m]
        return GTColumns(**kw)

    def to_json(self):
        """Inverse of from_json (used by the synthetic generator and tests)."""
        def num(x):
            x = float(x)
            return int(x) if x == int(x) else x
        merged_by = {}
        for src, dst in self.cat_merged.tolist():
            merged_by.setdefault(dst, []).append({"id": src})
        cats = []
        for cid, fr in zip(self.cat_id.tolist(), self.cat_freq.tolist()):
            c = {"id": cid, "name": "c%d" % cid, "frequency": chr(fr)}
            if cid in merged_by:
                c["merged"] = merged_by[cid]
            cats.append(c)
        vneg = _uncsr(self.vid_neg_off, self.vid_neg)
        vnel = _uncsr(self.vid_nel_off, self.vid_nel)
        vids = [{"id": v, "name": "v%d" % v, "neg_category_ids": vneg[k],
                 "not_exhaustive_category_ids": vnel[k]}
                for k, v in enumerate(self.vid_id.tolist())]
        ineg = _uncsr(self.img_neg_off, self.img_neg)
        inel = _uncsr(self.img_nel_off, self.img_nel)
        imgs = [{"id": i, "video_id": v, "frame_index": num(fi),
                 "neg_category_ids": ineg[k],
                 "not_exhaustive_category_ids": inel[k]}
                for k, (i, v, fi) in enumerate(zip(self.img_id.tolist(),
                                                   self.img_vid.tolist(),
                                                   self.img_frame.tolist()))]
        trks = []
        for t, c, v, ig in zip(self.trk_id.tolist(), self.trk_cat.tolist(),
                               self.trk_vid.tolist(),
                               self.trk_ignore.tolist()):
            d = {"id": t, "category_id": c, "video_id": v}
            if ig:
                d["ignore"] = 1
            trks.append(d)
        anns = []
        bb = self.ann_bbox.tolist()
        for k in range(len(self.ann_id)):
            d = {"id": int(self.ann_id[k]), "image_id": int(self.ann_img[k]),
                 "track_id": int(self.ann_trk[k]),
                 "category_id": int(self.ann_cat[k]),
                 "bbox": [num(x) for x in bb[k]],
                 "area": num(self.ann_area[k]),
                 "visibility": num(self.ann_vis[k]),
                 "out_of_frame": bool(self.ann_oof[k])}
            if self.ann_ignore[k]:
                d["ignore"] = 1
            anns.append(d)
        return {"info": {"description": "synthetic"}, "images": imgs,
                "videos": vids, "tracks": trks, "annotations": anns,
                "categories": cats}


def fingerprint(v):
    """A cheap digest of an array's CONTENT -- ~1000 evenly spaced elements and
    both ends -- so that the usual in-place edits of an uploaded column between
    two evaluations (scores rescaled, boxes shifted, ids remapped: ADVICE r3)
    are seen as well as a rebound column; an edit of single elements between
    the samples still needs flatten_dev.forget_columns(dt)."""
    if not isinstance(v, np.ndarray) or v.size == 0:
        return 0
    flat = v.reshape(-1) if v.flags.c_contiguous else np.ascontiguousarray(v).reshape(-1)
    step = max(1, flat.size // 1024)
    return hash((flat[::step].tobytes(), flat[-1:].tobytes()))


class DTColumns:
    """Prediction list as arrays (file order preserved)."""

    FIELDS = ("image_id", "category_id", "bbox", "score", "track_id",
              "video_id")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw[f])

    def __len__(self):
        return len(self.image_id)

    @classmethod
    def from_file_native(cls, path, part=0, n_parts=1):
        """Parse a prediction file with the native columnar reader
        (csrc/ingest.cpp); returns None when that library is not built.
        ``part`` of ``n_parts``: one process's share of the list (multi-process
        CLI) -- the result then carries ``first`` (position of its first record
        in the file) and ``total`` (records in the file)."""
        import ctypes as C
        import os
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                          "libtao_amodal_ingest.so")
        if not os.path.exists(so):
            return None
        lib = C.CDLL(so)
        if n_parts == 1:
            out = cls._from_file_device(path, lib)
            if out is not None:
                return out
        lib.taoamd_pred_scan.restype = C.c_void_p
        lib.taoamd_pred_scan.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_char_p,
                                         C.c_size_t]
        lib.taoamd_pred_scan_info.argtypes = [C.c_void_p] * 4
        lib.taoamd_pred_convert.argtypes = [C.c_void_p] * 7 + [C.c_char_p, C.c_size_t]
        lib.taoamd_pred_scan_free.argtypes = [C.c_void_p]
        err = C.create_string_buffer(512)

        def refuse(msg):
            if "is not a list" in msg:
                raise AssertionError("results is not a list.")
            if msg.startswith("cannot open"):
                raise FileNotFoundError(msg)
            if "KeyError: '" in msg:
                # a required key is absent: what the reference's dict access raises
                raise KeyError(msg.split("KeyError: '")[1].split("'")[0])
            raise ValueError("malformed prediction file: " + msg)
        h = lib.taoamd_pred_scan(os.fsencode(path), part, n_parts, err, 512)
        if not h:
            refuse(err.value.decode())
        converted = False
        try:
            first, n, total = C.c_int64(0), C.c_int64(0), C.c_int64(0)
            lib.taoamd_pred_scan_info(h, C.byref(first), C.byref(n), C.byref(total))
            n = n.value
            i64, f64 = np.int64, np.float64
            # (the numbers are converted straight into these arrays, every
            # core touching its own rows first)
            out = cls(image_id=np.empty(n, i64), category_id=np.empty(n, i64),
                      bbox=np.empty((n, 4), f64), score=np.empty(n, f64),
                      track_id=np.empty(n, i64), video_id=np.empty(n, i64))
            rc = lib.taoamd_pred_convert(
                h, out.image_id.ctypes.data, out.category_id.ctypes.data,
                out.bbox.ctypes.data, out.score.ctypes.data, out.track_id.ctypes.data,
                out.video_id.ctypes.data, err, 512)
            if rc:
                refuse(err.value.decode())
            out.first, out.total = first.value, total.value
            converted = True
        finally:
            if converted and n >= 1000000:
                # unmapping the file and releasing the element table (0.05 s at
                # 30 M predictions) is nobody's business but the allocator's:
                # off the caller's path
                import threading
                threading.Thread(target=lib.taoamd_pred_scan_free, args=(h,),
                                 daemon=True).start()
            else:
                lib.taoamd_pred_scan_free(h)
        return out

    # files from this size on are read on the device (csrc/json_ingest.hip)
    DEVICE_INGEST_MIN_BYTES = 32 << 20
    DEVICE_INGEST_FLAG_CAP = 4096

    @classmethod
    def _from_file_device(cls, path, host_lib):
        """The prediction file read on the GPU (taoamd_json_pred_open / _read:
        the text copied into HBM, the list's objects found and converted there).
        Returns None when the host reader should take the file: a small file,
        no GPU, ``TAOAMD_DEVICE_INGEST=0``, or a file holding anything the
        device reader leaves to the host's (a backslash, elements that are not
        objects, more unusual objects than it lists).  Objects it lists --
        literals, ids that are not plain integers, missing keys -- are read by
        the host reader's parse_object and patched in, with its errors."""
        import ctypes as C
        import sys
        if os.environ.get("TAOAMD_DEVICE_INGEST", "1") == "0":
            return None
        try:
            least = int(os.environ.get("TAOAMD_DEVICE_INGEST_MIN_BYTES",
                                       cls.DEVICE_INGEST_MIN_BYTES))
            if os.path.getsize(path) < least:
                return None
        except OSError:
            return None                 # (the host reader reports the missing file)
        from . import _lib
        fresh = EARLY_HIP.get("event") is not None and not EARLY_HIP.get("torch_loaded")
        if not fresh and "torch" not in sys.modules:
            return None                 # (no torch, nobody starting the runtime: the host reader)
        if fresh:
            # A fresh process whose torch is still being imported on another
            # thread: _lib.load() would wait for that import.  The drop-in CLI
            # starts the HIP runtime beside it (EARLY_HIP: torch's own copy of
            # the runtime, loaded first, so the kernel library binds to it) --
            # then the library is loaded here without torch and the columns come
            # back as host arrays; nobody started the runtime: the host reader.
            ev = EARLY_HIP.get("event")
            if ev is None:
                return None
            ev.wait()
            if not EARLY_HIP.get("ok") or not os.path.exists(_lib.SO_PATH):
                return None
            hip = C.CDLL(_lib.SO_PATH)
            for name in ("taoamd_json_pred_open", "taoamd_json_pred_workspace",
                         "taoamd_json_pred_count",
                         "taoamd_json_pred_read", "taoamd_json_pred_close",
                         "taoamd_last_error"):
                fn = getattr(hip, name)
                fn.restype, fn.argtypes = _lib.SIGNATURES[name]
        else:
            try:
                hip = _lib.load()
            except OSError:
                return None
        err = C.create_string_buffer(512)
        status = C.c_int32(0)
        work, work_ptr, work_bytes = None, None, 0
        if not fresh:
            # text, tables and object offsets in memory of torch's caching
            # allocator (the CLI's warm-up keeps a block of the file's size
            # there): a raw hipMalloc / hipFree of gigabytes takes 0.1 s and more
            # in a process whose allocator already holds most of what it uses
            import torch
            if not torch.cuda.is_available():
                return None             # (no GPU in this process: the host reader)
            try:
                work_bytes = int(hip.taoamd_json_pred_workspace(os.path.getsize(path)))
                work = torch.empty(work_bytes, dtype=torch.uint8,
                                   device=torch.device("cuda", torch.cuda.current_device()))
                torch.cuda.current_stream().synchronize()
                work_ptr = work.data_ptr()
            except (torch.cuda.OutOfMemoryError, OSError):
                work, work_ptr, work_bytes = None, None, 0
        h = hip.taoamd_json_pred_open(os.fsencode(path), work_ptr, work_bytes,
                                      C.byref(status), err, 512, None)
        if not h:
            if status.value == _lib.JSON_FALLBACK or status.value == _lib.ERR_ARG:
                return None
            raise RuntimeError("taoamd_json_pred_open: %s (%s)"
                               % (err.value.decode(), hip.taoamd_last_error().decode()))
        cap = cls.DEVICE_INGEST_FLAG_CAP
        flag, flag_at = np.zeros(cap, np.int64), np.zeros(cap, np.int64)
        n_flag = C.c_int32(0)
        out = t = dev = None
        try:
            n = hip.taoamd_json_pred_count(h)
            if fresh:
                i64, f64 = np.int64, np.float64
                out = cls(image_id=np.empty(n, i64), category_id=np.empty(n, i64),
                          bbox=np.empty((n, 4), f64), score=np.empty(n, f64),
                          track_id=np.empty(n, i64), video_id=np.empty(n, i64))
                rc = hip.taoamd_json_pred_read(
                    h, out.image_id.ctypes.data, out.category_id.ctypes.data,
                    out.bbox.ctypes.data, out.score.ctypes.data, out.track_id.ctypes.data,
                    out.video_id.ctypes.data, flag.ctypes.data, flag_at.ctypes.data, cap,
                    C.byref(n_flag))
                if rc:
                    raise RuntimeError("taoamd_json_pred_read: status %d (%s)"
                                       % (rc, hip.taoamd_last_error().decode()))
            else:
                import torch
                dev = torch.device("cuda", torch.cuda.current_device())
                i64, f64 = torch.int64, torch.float64
                with torch.cuda.device(dev):
                    try:
                        t = {"image_id": torch.empty(n, dtype=i64, device=dev),
                             "category_id": torch.empty(n, dtype=i64, device=dev),
                             "bbox": torch.empty((n, 4), dtype=f64, device=dev),
                             "score": torch.empty(n, dtype=f64, device=dev),
                             "track_id": torch.empty(n, dtype=i64, device=dev),
                             "video_id": torch.empty(n, dtype=i64, device=dev)}
                    except torch.cuda.OutOfMemoryError:
                        return None     # (no room for the columns: the host reader)
                    torch.cuda.current_stream().synchronize()
                    _lib.check(hip.taoamd_json_pred_convert(
                        h, t["image_id"].data_ptr(), t["category_id"].data_ptr(),
                        t["bbox"].data_ptr(), t["score"].data_ptr(),
                        t["track_id"].data_ptr(), t["video_id"].data_ptr(),
                        flag.ctypes.data, flag_at.ctypes.data, cap, C.byref(n_flag)),
                        "taoamd_json_pred_convert")
        finally:
            # (the text, the tables and the mapping are released off the
            # caller's path: 0.04 s at 30 M predictions)
            import threading

            def release(h=h, work=work):
                hip.taoamd_json_pred_close(h)
                del work                # (the workspace goes back to its pool after the handle)
            threading.Thread(target=release, daemon=True).start()
        if n_flag.value > cap:
            return None
        if n_flag.value == 0:
            if out is not None:
                out.first, out.total = 0, n
                return out
            # the columns stay where they were made: the table builds read
            # them there, the host arrays arrive in the background
            return DeviceDTColumns(n, t, dev)
        # some objects are the host reader's: plain host columns, patched
        if out is None:
            out = cls(**{f: t[f].cpu().numpy() for f in cls.FIELDS})
            del t
        k = n_flag.value
        order = np.argsort(flag[:k], kind="stable")
        idx = np.ascontiguousarray(flag[:k][order])
        at = np.ascontiguousarray(flag_at[:k][order])
        host_lib.taoamd_pred_patch.argtypes = [C.c_char_p, C.c_int64] + [C.c_void_p] * 8 \
            + [C.c_char_p, C.c_size_t]
        rc = host_lib.taoamd_pred_patch(
            os.fsencode(path), k, idx.ctypes.data, at.ctypes.data,
            out.image_id.ctypes.data, out.category_id.ctypes.data, out.bbox.ctypes.data,
            out.score.ctypes.data, out.track_id.ctypes.data, out.video_id.ctypes.data,
            err, 512)
        if rc:
            # (the host reader scans the file again and raises what it finds:
            # its message for the list's first malformed record)
            return None
        out.first, out.total = 0, n
        return out

    @classmethod
    def from_json(cls, results):
        if isinstance(results, str):
            native = cls.from_file_native(results)
            if native is not None:
                return native
            with open(results, "r") as f:
                results = json.load(f)
        assert isinstance(results, list), "results is not a list."
        i64 = np.int64
        return cls(
            image_id=np.asarray([r["image_id"] for r in results], dtype=i64),
            category_id=np.asarray([r["category_id"] for r in results],
                                   dtype=i64),
            bbox=np.asarray([r["bbox"] for r in results],
                            dtype=np.float64).reshape(-1, 4),
            score=np.asarray([r["score"] for r in results], dtype=np.float64),
            track_id=np.asarray([r.get("track_id", -1) for r in results],
                                dtype=i64),
            video_id=np.asarray([r.get("video_id", -1) for r in results],
                                dtype=i64),
        )

    def to_json(self):
        def num(x):
            x = float(x)
            return int(x) if x == int(x) else x
        bb = self.bbox.tolist()
        return [{"image_id": int(self.image_id[k]),
                 "category_id": int(self.category_id[k]),
                 "bbox": [num(x) for x in bb[k]],
                 "score": float(self.score[k]),
                 "track_id": int(self.track_id[k]),
                 "video_id": int(self.video_id[k])}
                for k in range(len(self.image_id))]

    def take(self, idx):
        out = DTColumns(**{f: getattr(self, f)[idx] for f in self.FIELDS})
        # (per-row extras ride along: a share's place in the file's list,
        # evaluation/_dist.shard_inputs)
        if getattr(self, "file_pos", None) is not None:
            out.file_pos = np.asarray(self.file_pos)[idx]
        return out

    def write_json(self, path):
        """The prediction list written by the native writer (csrc/
        jsonwrite.cpp)."""
        lib = _ingest_lib()
        i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        cols = [i64(self.image_id), i64(self.category_id),
                np.ascontiguousarray(self.bbox, dtype=np.float64),
                np.ascontiguousarray(self.score, dtype=np.float64),
                i64(self.track_id), i64(self.video_id)]
        rc = lib.taoamd_pred_write(os.fsencode(path), len(cols[0]),
                                   *[c.ctypes.data for c in cols])
        if rc:
            raise OSError("taoamd_pred_write(%s) failed: %d" % (path, rc))

    @classmethod
    def concat(cls, parts):
        return cls(**{f: np.concatenate([getattr(p, f) for p in parts])
                      for f in cls.FIELDS})


class DeviceDTColumns(DTColumns):
    """Prediction columns made ON THE DEVICE (DTColumns._from_file_device): the
    device tensors are there at once -- flatten_dev.raw_columns hands them to
    the table builds instead of uploading host arrays -- and the host arrays
    every other reader of a DTColumns sees arrive in the background, column by
    column (image_id, track_id, video_id first: what the constructors of the
    Results classes check on the host).  Reading a column waits for its copy;
    assigning one (``dt.score = other``) replaces it and retires the device
    copy, and an in-place edit of an arrived array is seen by its fingerprint,
    as for uploaded columns."""

    ORDER = ("image_id", "track_id", "video_id", "category_id", "score", "bbox")

    def __init__(self, n, tensors, device):
        import threading
        d = self.__dict__
        d["_n"], d["_dev"], d["_device"] = n, dict(tensors), device
        d["_host"], d["_fp"], d["_error"] = {}, {}, None
        d["_ready"] = {f: threading.Event() for f in self.FIELDS}
        d["first"], d["total"] = 0, n
        threading.Thread(target=self._download, daemon=True).start()

    def _download(self):
        try:
            import torch
            with torch.cuda.device(self._device):
                side = torch.cuda.Stream(device=self._device)
                with torch.cuda.stream(side):
                    for f in self.ORDER:
                        t = self._dev.get(f)
                        if self._ready[f].is_set() or t is None:
                            continue            # (assigned meanwhile)
                        arr = np.empty(tuple(t.shape), dtype=np.float64
                                       if t.dtype == torch.float64 else np.int64)
                        torch.from_numpy(arr).copy_(t)
                        side.synchronize()
                        if not self._ready[f].is_set():
                            self._fp[f] = fingerprint(arr)
                            self._host[f] = arr
                            self._ready[f].set()
        except BaseException as e:      # (a reader of the columns re-raises it)
            self.__dict__["_error"] = e
            for ev in self._ready.values():
                ev.set()

    def __len__(self):
        return self._n

    def _get(self, f):
        self._ready[f].wait()
        if self._error is not None and f not in self._host:
            raise RuntimeError("prediction columns could not be copied from the "
                               "device") from self._error
        return self._host[f]

    def _set(self, f, v):
        if self._ready[f].is_set() and v is self._host.get(f):
            return                      # (the same array handed back)
        self._host[f] = v
        self._dev.pop(f, None)
        self._fp.pop(f, None)
        self._ready[f].set()

    def track_clash_free(self):
        """Whether every track id occurs with ONE video id (what
        make_track_ids_unique asks first, tools/eval_on_tao_amodal.py:44-58),
        answered where the columns were made: a table of the id's video, every
        row against it.  None: not answerable here (a column replaced, ids
        negative or in too wide a range) -- the host statement decides."""
        import torch
        cols = self.device_columns(self._device, ("track_id", "video_id"))
        if cols is None or self._n == 0:
            return None
        tid, vid = cols["track_id"], cols["video_id"]
        with torch.cuda.device(self._device):
            lo, hi = int(tid.min()), int(tid.max())
            if lo < 0 or hi >= (1 << 26):
                return None
            table = torch.empty(hi + 1, dtype=torch.int64, device=self._device)
            table[tid] = vid                    # (some row's video per id)
            return not bool((table[tid] != vid).any())

    def device_columns(self, device, names):
        """{name: tensor} of the columns as they were made, when they still
        are what the host arrays hold (None: upload the host arrays)."""
        import torch
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if device.type != "cuda" or idx != self._device.index:
            return None
        out = {}
        for f in names:
            t = self._dev.get(f)
            if t is None:
                return None
            if self._ready[f].is_set() and (self._error is not None
                                            or fingerprint(self._host[f]) != self._fp.get(f)):
                self._dev.pop(f, None)
                return None
            out[f] = t
        return out


for _f in DTColumns.FIELDS:
    setattr(DeviceDTColumns, _f, property(lambda self, _f=_f: self._get(_f),
                                          lambda self, v, _f=_f: self._set(_f, v)))
del _f
