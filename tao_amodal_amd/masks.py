"""Run-length masks for ``LVISEval(iou_type="segm")``: the host side.

``MaskBatch.add(segm, h, w)`` takes what ``LVIS.ann_to_rle`` takes (reference
lvis_amodal/lvis.py:171-193) -- a polygon list, an uncompressed RLE or a
compressed RLE -- and appends the mask to a native batch
(``csrc/rle.cpp`` in libtao_amodal_ingest.so: rasterisation, union, text
form); ``arrays()`` returns the back-to-back run lengths the device kernel
``taoamd_rle_iou`` walks, with areas and tight boxes (``mask_utils.area`` /
``toBbox``, lvis_amodal/results.py:54-60).  No Python or numpy fallback: the
library has to be built (csrc/build.sh).
"""
import ctypes as C
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                   "libtao_amodal_ingest.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(
                "native host library not built: %s is missing (run "
                "tao_amodal_amd/csrc/build.sh); run-length masks have no "
                "Python fallback" % _SO)
        lib = C.CDLL(_SO)
        vp, i64 = C.c_void_p, C.c_int64
        lib.taoamd_rle_new.restype = vp
        lib.taoamd_rle_free.argtypes = [vp]
        lib.taoamd_rle_count.restype = i64
        lib.taoamd_rle_count.argtypes = [vp]
        lib.taoamd_rle_total.restype = i64
        lib.taoamd_rle_total.argtypes = [vp]
        lib.taoamd_rle_add_polygons.restype = i64
        lib.taoamd_rle_add_polygons.argtypes = [vp, C.c_int32, vp, vp, i64, i64]
        lib.taoamd_rle_add_polygon_batch.restype = i64
        lib.taoamd_rle_add_polygon_batch.argtypes = [vp, i64, vp, vp, vp, vp]
        lib.taoamd_rle_add_counts.restype = i64
        lib.taoamd_rle_add_counts.argtypes = [vp, vp, i64, i64, i64]
        lib.taoamd_rle_add_string.restype = i64
        lib.taoamd_rle_add_string.argtypes = [vp, C.c_char_p, i64, i64]
        lib.taoamd_rle_copy.restype = C.c_int
        lib.taoamd_rle_copy.argtypes = [vp, vp, vp, vp, vp, vp]
        lib.taoamd_rle_string.restype = i64
        lib.taoamd_rle_string.argtypes = [vp, i64, vp, i64]
        _lib = lib
    return _lib


class MaskArrays:
    """CSR run lengths of n masks: off[n+1] (int64), counts (uint32),
    hw[n, 2] (int32: height, width), area[n] (uint32), bbox[n, 4] (float64)."""

    def __init__(self, off, counts, hw, area, bbox):
        self.off, self.counts, self.hw = off, counts, hw
        self.area, self.bbox = area, bbox

    def __len__(self):
        return len(self.off) - 1

    def slice(self, a, b):
        """Masks [a, b) as their own CSR."""
        lo, hi = int(self.off[a]), int(self.off[b])
        return MaskArrays((self.off[a:b + 1] - lo).astype(np.int64),
                          self.counts[lo:hi], self.hw[a:b], self.area[a:b],
                          self.bbox[a:b])

    def mask(self, i):
        """Mask i as the oracle's dict (tests)."""
        return {"h": int(self.hw[i, 0]), "w": int(self.hw[i, 1]),
                "counts": self.counts[self.off[i]:self.off[i + 1]].tolist()}


class MaskBatch:
    def __init__(self):
        self.lib = _load()
        self.h = self.lib.taoamd_rle_new()

    def close(self):
        if self.h:
            self.lib.taoamd_rle_free(self.h)
            self.h = None

    __del__ = close

    def __len__(self):
        return int(self.lib.taoamd_rle_count(self.h))

    def add(self, segm, height, width):
        """Append the mask of one annotation; returns its index."""
        lib = self.lib
        if isinstance(segm, list):
            # frPyObjects looks at the first element only (_mask.pyx:277-296):
            # four numbers make it a list of boxes, which frBbox then rejects
            # because it is typed as an ndarray
            if len(segm[0]) == 4:
                raise TypeError("Argument 'bb' has incorrect type (expected "
                                "numpy.ndarray, got list)")
            if not len(segm[0]) > 4:
                raise Exception("input type is not supported.")
            parts = [np.asarray(p, dtype=np.float64).ravel() for p in segm]
            off = np.zeros(len(parts) + 1, dtype=np.int64)
            np.cumsum([len(p) for p in parts], out=off[1:])
            xy = np.ascontiguousarray(np.concatenate(parts))
            k = lib.taoamd_rle_add_polygons(self.h, len(parts), off.ctypes.data,
                                            xy.ctypes.data, int(height), int(width))
        elif isinstance(segm["counts"], list):
            c = np.ascontiguousarray(np.array(segm["counts"], dtype=np.uint32))
            k = lib.taoamd_rle_add_counts(self.h, c.ctypes.data, len(c),
                                          int(segm["size"][0]), int(segm["size"][1]))
        else:
            text = segm["counts"]
            if isinstance(text, str):
                text = text.encode()
            k = lib.taoamd_rle_add_string(self.h, text, int(segm["size"][0]),
                                          int(segm["size"][1]))
        if k < 0:
            raise ValueError("cannot make a run-length mask of %r" % (segm,))
        return int(k)

    def add_many(self, items):
        """Append the masks of many annotations: items = iterable of (segm,
        height, width).  Runs of polygon annotations go to the library in one
        call (rasterised on all cores); the other forms one by one.  Same
        results and order as add() in a loop."""
        lib = self.lib
        xy, part_off, mask_part_off, hw = [], [0], [0], []

        def flush():
            if not hw:
                return
            a_xy = np.asarray(xy, dtype=np.float64)
            a_po = np.asarray(part_off, dtype=np.int64)
            a_mo = np.asarray(mask_part_off, dtype=np.int64)
            a_hw = np.asarray(hw, dtype=np.int32).reshape(-1, 2)
            k = lib.taoamd_rle_add_polygon_batch(
                self.h, len(a_hw), a_mo.ctypes.data, a_po.ctypes.data,
                a_xy.ctypes.data, a_hw.ctypes.data)
            if k < 0:
                raise ValueError("cannot make run-length masks of a polygon batch")
            del xy[:], part_off[1:], mask_part_off[1:], hw[:]

        for segm, height, width in items:
            if isinstance(segm, list) and len(segm[0]) > 4:
                for p in segm:
                    xy.extend(p)
                    part_off.append(len(xy))
                mask_part_off.append(len(part_off) - 1)
                hw.append((int(height), int(width)))
            else:
                flush()
                self.add(segm, height, width)     # (also raises for bad input)
        flush()

    def arrays(self):
        n, tot = len(self), int(self.lib.taoamd_rle_total(self.h))
        off = np.zeros(n + 1, dtype=np.int64)
        counts = np.zeros(tot, dtype=np.uint32)
        hw = np.zeros((n, 2), dtype=np.int32)
        area = np.zeros(n, dtype=np.uint32)
        bbox = np.zeros((n, 4), dtype=np.float64)
        self.lib.taoamd_rle_copy(self.h, off.ctypes.data, counts.ctypes.data,
                                 hw.ctypes.data, area.ctypes.data, bbox.ctypes.data)
        return MaskArrays(off, counts, hw, area, bbox)

    def text(self, i):
        """Compressed text of mask i (what the reference keeps in
        ann["segmentation"]["counts"] after _to_mask, as bytes there)."""
        n = int(self.lib.taoamd_rle_string(self.h, int(i), None, 0))
        if n < 0:
            raise IndexError(i)
        buf = C.create_string_buffer(n + 1)
        self.lib.taoamd_rle_string(self.h, int(i), buf, n + 1)
        return buf.value.decode("ascii")
