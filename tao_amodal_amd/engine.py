"""Device pipeline: flattened cell tables -> precision / recall on the GPU.

PyTorch is used only as the allocator/stream provider (device tensors,
``data_ptr()``, ``torch.cuda.current_stream()``); every arithmetic step is a
hand-written HIP kernel behind the C ABI of ``include/tao_amodal_hip.h``.

One evaluator pass ("step" of bench.py) =

    ranges  ->  sort (category, -score)  ->  [track 3D IoU]  ->
    IoU + greedy match (writes rows in sorted order)  ->  accumulate

which covers reference LVISEval.evaluate()+accumulate() (lvis_amodal/eval.py:
115-145,305-426) or TaoEval.evaluate()+accumulate() (tao_amodal/eval.py:
246-276,459-584) for the non-empty cells.
"""
import os

import numpy as np
import torch

from . import _lib

N_THR, N_REC = _lib.N_THR, _lib.N_REC


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


_RAW_TABLES = {}     # id(host table) -> (weakref, {device: tensor})
import threading as _threading
_RAW_TABLES_LOCK = _threading.Lock()


def _to_device(v, device):
    """Upload a table of a flattened problem.  flatten.LazyRows (rows of a raw
    input table) are gathered on the device from ONE upload of the raw table,
    shared by every problem built from the same input."""
    from .flatten import LazyRows
    if not isinstance(v, LazyRows):
        return torch.from_numpy(np.ascontiguousarray(v)).to(device)
    import weakref
    key = id(v.source)
    dev = torch.device(device)
    with _RAW_TABLES_LOCK:
        ent = _RAW_TABLES.get(key)
        if ent is None or ent[0]() is not v.source:
            ent = (weakref.ref(v.source, lambda _r, k=key: _RAW_TABLES.pop(k, None)), {})
            _RAW_TABLES[key] = ent
        if dev not in ent[1]:
            # (under the lock: the CLI builds both levels' problems side by side)
            ent[1][dev] = torch.from_numpy(np.ascontiguousarray(v.source)).to(dev)
    if len(v.index) == 0:
        return torch.zeros((0,) + tuple(v.source.shape[1:]), dtype=ent[1][dev].dtype,
                           device=dev)
    return ent[1][dev].index_select(0, torch.from_numpy(v.index).to(dev))


def match_plan(cell_dt_off, cell_gt_off, cap_d=64, cap_g=64, cap_cell_g=8):
    """Pack runs of consecutive small cells (<= cap_d detections and <= cap_g
    GTs in total, <= cap_cell_g GTs per cell) into groups for
    match_group_kernel; every other cell that has detections is a 'single'
    (taoamd_match_plan_host)."""
    lib = _lib.load()
    d_off = np.ascontiguousarray(cell_dt_off, dtype=np.int32)
    g_off = np.ascontiguousarray(cell_gt_off, dtype=np.int32)
    sizes = np.zeros(2, dtype=np.int64)
    args = (len(d_off) - 1, d_off.ctypes.data, g_off.ctypes.data, cap_d, cap_g,
            cap_cell_g, sizes.ctypes.data)
    _lib.check(lib.taoamd_match_plan_host(*args, None, None),
               "taoamd_match_plan_host")
    ng, ns = int(sizes[0]), int(sizes[1])
    groups = np.zeros((max(ng, 1), 2), dtype=np.int32)
    singles = np.zeros(max(ns, 1), dtype=np.int32)
    if ng or ns:
        _lib.check(lib.taoamd_match_plan_host(
            *args, groups.ctypes.data, singles.ctypes.data),
            "taoamd_match_plan_host")
    return groups[:ng], singles[:ns]


def sort_plan(cat_off):
    """Chunks / buckets / scatter tiles of taoamd_sort_sampled from the host
    copy of the category offsets (taoamd_sort_plan_host).  Returns the four
    int32 tables and whether some category needs the merge passes."""
    lib = _lib.load()
    off = np.ascontiguousarray(cat_off, dtype=np.int32)
    sizes = np.zeros(5, dtype=np.int64)
    _lib.check(lib.taoamd_sort_plan_host(len(off) - 1, off.ctypes.data,
                                         sizes.ctypes.data, None, None, None, None),
               "taoamd_sort_plan_host")
    nc, ns, nt, nb = (int(x) for x in sizes[:4])
    chunks = np.zeros((max(nc, 1), 8), dtype=np.int32)
    split = np.zeros(max(ns, 1), dtype=np.int32)
    stile = np.zeros(max(nt, 1), dtype=np.int32)
    bucket = np.zeros(max(nb, 1), dtype=np.int32)
    if nc:
        _lib.check(lib.taoamd_sort_plan_host(
            len(off) - 1, off.ctypes.data, sizes.ctypes.data, chunks.ctypes.data,
            split.ctypes.data, stile.ctypes.data, bucket.ctypes.data),
            "taoamd_sort_plan_host")
    return (chunks, split, stile, bucket), (nc, ns, nt, nb), bool(sizes[4])


def track_meta(flat):
    """{first, last, base - first, is detection} of every track (detection
    tracks first, then GT) for the padded frame table: track t owns the slots
    base .. base + last - first; slot 0 is the table's far box.  Returns the
    int32 table, the slot range of either side and the table size."""
    metas, sides, n_slots = [], {}, 1
    for side in ("dt", "gt"):
        dev = getattr(flat, "dev", {})
        if side + "_frame_off" in dev and side + "_frame_pos" in dev \
                and not dict.__contains__(flat, side + "_frame_pos"):
            # device-built tables: the tracks' first / last positions are
            # gathered where the frame lists are (21 M positions at the
            # validation scale: 0.04 s to bring them to the host for this)
            off_t, pos_t = dev[side + "_frame_off"].long(), dev[side + "_frame_pos"]
            has_t = off_t[1:] > off_t[:-1]
            top = max(int(pos_t.numel()) - 1, 0)
            if pos_t.numel():
                first_t = torch.where(has_t, pos_t[off_t[:-1].clamp(max=top)].long(),
                                      torch.ones_like(off_t[1:]))
                last_t = torch.where(has_t, pos_t[(off_t[1:] - 1).clamp(min=0, max=top)].long(),
                                     torch.zeros_like(off_t[1:]))
            else:
                first_t, last_t = torch.ones_like(off_t[1:]), torch.zeros_like(off_t[1:])
            both = torch.stack([first_t, last_t]).cpu().numpy()
            first, last = both[0], both[1]
            has = last >= first
        else:
            off = np.asarray(flat[side + "_frame_off"], dtype=np.int64)
            pos = np.asarray(flat[side + "_frame_pos"], dtype=np.int64)
            has = off[1:] > off[:-1]
            first = np.ones(len(off) - 1, dtype=np.int64)
            last = np.zeros(len(off) - 1, dtype=np.int64)
            first[has] = pos[off[:-1][has]]
            last[has] = pos[off[1:][has] - 1]
        span = np.where(has, last - first + 1, 0)
        base = n_slots + np.cumsum(span) - span
        metas.append(np.stack([first, last, base - first,
                               np.full_like(first, side == "dt")], 1))
        total = int(span.sum())
        # the first set also fills slot 0
        sides[side] = (n_slots - (side == "dt"), total + (side == "dt"))
        n_slots += total
    return np.concatenate(metas), sides, n_slots


def track_iou_plan(flat, meta):
    """Launch plan of taoamd_track_iou_planned (one wavefront per task: up to
    32 tracks, up to 64 track pairs).  Returns tasks[n, 4], task_rows,
    task_pairs (int32) and task_out (int64)."""
    import ctypes as C
    lib = _lib.load()
    d_off = np.ascontiguousarray(flat.cell_dt_off, dtype=np.int32)
    g_off = np.ascontiguousarray(flat.cell_gt_off, dtype=np.int32)
    i_off = np.ascontiguousarray(flat.cell_iou_off, dtype=np.int64)
    meta = np.ascontiguousarray(meta, dtype=np.int32)
    sizes = np.zeros(3, dtype=np.int64)
    args = (len(d_off) - 1, d_off.ctypes.data, g_off.ctypes.data,
            i_off.ctypes.data, meta.ctypes.data, sizes.ctypes.data)
    _lib.check(lib.taoamd_track_iou_plan_host(*args, None, None, None, None),
               "taoamd_track_iou_plan_host")
    tasks = np.zeros((int(sizes[0]), 4), dtype=np.int32)
    rows = np.zeros(int(sizes[1]), dtype=np.int32)
    pairs = np.zeros(int(sizes[2]), dtype=np.int32)
    out = np.zeros(int(sizes[2]), dtype=np.int64)
    if sizes[0]:
        _lib.check(lib.taoamd_track_iou_plan_host(
            *args, tasks.ctypes.data, rows.ctypes.data, pairs.ctypes.data,
            out.ctypes.data), "taoamd_track_iou_plan_host")
    return tasks, rows, pairs, out


class DeviceProblem:
    """A flattened problem (flatten.Flat) resident in HBM."""

    IOU_MODES = {"3d_iou": 0, "avg_iou": 1, "imagenetvid": 2}

    def __init__(self, flat, device="cuda", iou_3d_type="3d_iou", guard="device"):
        """`guard`: where the frame-order guard recomputes the listed pairs --
        "device" (taoamd_track_iou_setorder, no host round trip; falls back to
        the host when an image id is outside [0, 2^61 - 1)) or "host" (Python's
        own sets, one synchronisation per pass: the cross-check of the tests)."""
        self.kind = flat.kind
        self.iou_mode = self.IOU_MODES[iou_3d_type]
        self.device = torch.device(device)
        self.n_rng = _lib.LVIS_RNG if self.kind == "lvis" else _lib.TAO_RNG
        self.n_words = (self.n_rng * N_THR + 63) // 64
        self.n_cells = int(flat.n_cells)
        self.n_cat = len(flat.cat_ids)
        self.n_dt = int(flat.cell_dt_off[-1])
        self.n_gt = int(flat.cell_gt_off[-1])
        self.n_pairs = int(flat.n_pairs)
        # tables a device-side build (flatten_dev.DeviceFlat) already holds in HBM
        on_dev = getattr(flat, "dev", {})
        d_cnt = np.diff(flat.cell_dt_off).astype(np.int64)
        g_cnt = np.diff(flat.cell_gt_off).astype(np.int64)
        self.max_g = int(g_cnt.max()) if self.n_cells else 0
        if self.max_g > _lib.MAX_GT_PER_CELL:
            raise _lib.TaoAmdError(
                "a cell holds %d ground truths; the kernels support up to %d"
                % (self.max_g, _lib.MAX_GT_PER_CELL))
        iou_off = np.zeros(self.n_cells + 1, dtype=np.int64)
        np.cumsum(d_cnt * g_cnt, out=iou_off[1:])
        self.n_iou = int(iou_off[-1])
        # category-major cells (flatten.py): every category is one run of
        # cells, of detections and of ground truths
        cell_cat = np.asarray(flat.cell_cat)
        self.grouped = bool(np.all(np.diff(cell_cat) >= 0))
        if self.grouped:
            first = np.searchsorted(cell_cat, np.arange(self.n_cat + 1), "left")
            cat_off = np.asarray(flat.cell_dt_off)[first].astype(np.int32)
            gt_cat_off = np.asarray(flat.cell_gt_off)[first].astype(np.int32)
        else:
            cat_off = np.zeros(self.n_cat + 1, dtype=np.int32)
            np.cumsum(np.bincount(flat.dt_cat, minlength=self.n_cat),
                      out=cat_off[1:])
            gt_cat_off = np.zeros(self.n_cat + 1, dtype=np.int32)
            np.cumsum(np.bincount(flat.gt_cat, minlength=self.n_cat),
                      out=gt_cat_off[1:])
        groups, singles = match_plan(flat.cell_dt_off, flat.cell_gt_off)
        self.n_groups, self.n_singles = len(groups), len(singles)
        names = ["cell_dt_off", "cell_gt_off", "dt_score", "dt_flags",
                 "dt_cat", "gt_flags", "gt_cat", "dt_cell"]
        if self.kind == "lvis":
            names += ["dt_box", "gt_box", "gt_vis"]
        else:
            names += ["dt_area", "dt_len", "gt_area", "gt_len", "gt_nhp",
                      "dt_frame_off", "dt_frame_pos", "dt_frame_box",
                      "gt_frame_off", "gt_frame_pos", "gt_frame_box"]
        self.t = {}
        for n in names:
            self.t[n] = on_dev[n].to(self.device) if n in on_dev else \
                _to_device(flat[n], self.device)
        # iou_type="segm": run-length masks, row i = detection / ground truth
        # i of the tables (masks.MaskArrays); the IoU then comes from
        # taoamd_rle_iou instead of the boxes
        masks = flat.get("masks") if self.kind == "lvis" else None
        self.mask_iou = masks is not None
        self.rle_total = {}
        if self.mask_iou:
            self.rle_total = {k: int(masks[k].off[-1]) for k in ("dt", "gt")}
            for side, n_rows in (("dt", self.n_dt), ("gt", self.n_gt)):
                m = masks[side]
                if len(m) != n_rows:
                    raise _lib.TaoAmdError(
                        "%d %s masks for %d rows" % (len(m), side, n_rows))
                self.t[side + "_rle_off"] = torch.from_numpy(m.off).to(self.device)
                self.t[side + "_rle_runs"] = torch.from_numpy(
                    np.ascontiguousarray(m.counts if len(m.counts) else
                                         np.zeros(1, np.uint32)).view(np.int32)
                ).to(self.device)
                self.t[side + "_rle_hw"] = torch.from_numpy(
                    np.ascontiguousarray(m.hw).reshape(-1, 2) if len(m) else
                    np.zeros((1, 2), np.int32)).to(self.device)
                self.t[side + "_rle_bb"] = torch.from_numpy(
                    np.ascontiguousarray(m.bbox).reshape(-1, 4) if len(m) else
                    np.zeros((1, 4))).to(self.device)
        self.max_segment = int(np.diff(cat_off).max()) if self.n_cat else 0
        # hint for the sweep: longest category (0 = unknown -> chunked kernels)
        self.acc_hint = self.max_segment
        tiles = (np.diff(cat_off) + _lib.SEGMENT_TILE - 1) // _lib.SEGMENT_TILE
        tile_off = np.zeros(self.n_cat + 1, dtype=np.int32)
        np.cumsum(tiles, out=tile_off[1:])
        self.n_tiles = int(tile_off[-1])
        # run descriptors {first detection, detections, first GT, GTs}
        runs = np.zeros((max(len(groups), 1), 4), dtype=np.int32)
        if len(groups):
            c0, c1 = groups[:, 0], groups[:, 1]
            runs[:, 0] = flat.cell_dt_off[c0]
            runs[:, 1] = flat.cell_dt_off[c1] - flat.cell_dt_off[c0]
            runs[:, 2] = flat.cell_gt_off[c0]
            runs[:, 3] = flat.cell_gt_off[c1] - flat.cell_gt_off[c0]
        self.t["groups"] = torch.from_numpy(runs).to(self.device)
        # (host copies: the multi-GPU plan cuts the launch plan into phases)
        self.groups_host = runs[:len(groups)]
        self.singles_host = np.asarray(singles, dtype=np.int32)
        self.singles_first_dt = np.asarray(flat.cell_dt_off)[self.singles_host] \
            if len(singles) else np.zeros(0, np.int32)
        self.n_tasks = 0
        self.single_frame = False
        self.exact_terms = False
        self.guard_flat = None         # host tables of the host-side guard
        self.guard_on_device = False
        self.near_ulp = 0
        if self.kind == "tao":
            self._plan_track_iou(flat)
            self._plan_guard(flat, guard)
        # per detection {first GT of its cell, GT count, position in the cell,
        # cell}: resolved here, on the device, from the uploaded cell tables
        cell = self.t["dt_cell"].long()
        g_off, d_off = self.t["cell_gt_off"], self.t["cell_dt_off"]
        self.t["dt_group"] = torch.stack(
            [g_off[cell], g_off[cell + 1] - g_off[cell],
             torch.arange(self.n_dt, dtype=torch.int32, device=self.device)
             - d_off[cell], self.t["dt_cell"]], dim=1).to(torch.int32).contiguous() \
            if self.n_dt else torch.zeros((1, 4), dtype=torch.int32,
                                          device=self.device)
        # image level: the same (relative to the run's first GT) and the flag
        # byte packed into one word per detection, all the group kernel reads
        # when it computes the IoUs itself (taoamd_match: dt_meta)
        self.t["dt_meta"] = None
        if self.kind == "lvis" and not self.mask_iou and self.n_dt and len(groups):
            dg, runs_t = self.t["dt_group"], self.t["groups"]
            d = torch.arange(self.n_dt, dtype=torch.int32, device=self.device)
            rid = (torch.searchsorted(runs_t[:, 0].contiguous(), d, right=True) - 1
                   ).clamp_(min=0)
            self.t["dt_meta"] = (
                self.t["dt_flags"].to(torch.int32)
                | (((dg[:, 0] - runs_t[:, 2][rid]) & 63) << 8)
                | (dg[:, 1].clamp(max=15) << 14) | ((dg[:, 2] & 63) << 18)).contiguous()
        self.t["singles"] = torch.from_numpy(
            singles if len(singles) else np.zeros(1, np.int32)).to(self.device)
        self.t["cell_iou_off"] = torch.from_numpy(iou_off).to(self.device)
        self.t["cat_off"] = torch.from_numpy(cat_off).to(self.device)
        self.t["gt_cat_off"] = torch.from_numpy(gt_cat_off).to(self.device)
        self.t["tile_off"] = torch.from_numpy(tile_off).to(self.device)
        self.cat_off_host = cat_off
        # plan of the sample sort (depends on cat_off alone)
        self.ss_sizes, self.ss_merge = (0, 0, 0, 0), False
        if self.grouped and self.n_dt:
            tabs, self.ss_sizes, self.ss_merge = sort_plan(cat_off)
            for name, tab in zip(("ss_chunks", "ss_split", "ss_stile", "ss_bucket"), tabs):
                self.t[name] = torch.from_numpy(tab).to(self.device)

    def _plan_track_iou(self, flat):
        """Padded frame table + launch plan of taoamd_track_iou_planned."""
        lib = _lib.load()
        for k in ("tasks", "task_rows", "task_pairs", "task_out", "frames",
                  "task_base", "trk_meta"):
            self.t[k] = None
        if self.device.type != "cuda":     # host-side plumbing tests: no kernels
            return
        meta, sides, n_slots = track_meta(flat)
        n_frames = sum(int(getattr(flat, "dev", {})[k].numel())
                       if k in getattr(flat, "dev", {}) and not dict.__contains__(flat, k)
                       else len(flat[k]) for k in ("dt_frame_pos", "gt_frame_pos"))
        self.n_slots = n_slots
        # tracks that are mostly holes would blow the table up: such inputs
        # take the two-pointer merge kernel instead (no plan)
        if n_slots > 4 * n_frames + (1 << 20) or n_slots >= 2 ** 31 - 1:
            return
        # every track a single frame (the stress shape: 10 000 one-frame
        # videos): a pair is ONE box IoU or 0 -- no sum, no order of summation,
        # nothing to guard; taoamd_track_iou_single instead of the task kernel,
        # whose LDS pipeline would be all prologue (1.05 ms for 2.9 M tracks)
        self.single_frame = (len(meta) > 0 and n_frames == len(meta)
                             and bool((meta[:, 0] == meta[:, 1]).all())
                             and os.environ.get("TAOAMD_SINGLE_FRAME", "1") != "0")
        if self.single_frame:
            self.exact_terms = True
            return
        tasks, rows, pairs, out = track_iou_plan(flat, meta)
        self.n_tasks = len(tasks)
        if self.n_tasks == 0:
            return
        dev = self.device
        self.t["tasks"] = torch.from_numpy(tasks).to(dev)
        self.t["task_rows"] = torch.from_numpy(rows).to(dev)
        self.t["task_pairs"] = torch.from_numpy(pairs).to(dev)
        self.t["task_out"] = torch.from_numpy(out).to(dev)
        self.t["trk_meta"] = torch.from_numpy(
            np.ascontiguousarray(meta, dtype=np.int32)).to(dev)
        padded = torch.empty((n_slots, 4), dtype=torch.float64, device=dev)
        # (sizes from the device tensors: a device-built table would bring its
        # frame lists to the host to answer len())
        n_dt = int(self.t["dt_frame_off"].numel()) - 1
        inexact = torch.zeros(1, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            for side, row0 in (("dt", 0), ("gt", n_dt)):
                slot0, slots = sides[side]
                n_trk = int(self.t[side + "_frame_off"].numel()) - 1
                _lib.check(lib.taoamd_track_pad(
                    n_trk, int(self.t[side + "_frame_pos"].numel()),
                    _ptr(self.t[side + "_frame_off"]),
                    _ptr(self.t[side + "_frame_pos"]),
                    _ptr(self.t[side + "_frame_box"]),
                    self.t["trk_meta"].data_ptr() + 16 * row0, slot0, slots,
                    _ptr(padded), _ptr(inexact), _stream()),
                    "taoamd_track_pad")
            # the frames in the order the tasks read them (taoamd_track_stream):
            # a row owns a 256-byte piece in every 8-position chunk its span
            # reaches into; a task's pieces lie chunk by chunk, rows in order
            has = meta[:, 1] >= meta[:, 0]
            trk_pieces = np.where(has, (meta[:, 1] >> 3) - (meta[:, 0] >> 3) + 1, 0)
            task_pieces = np.add.reduceat(trk_pieces[rows].astype(np.int64),
                                          tasks[:, 0].astype(np.int64))
            base = 1 + np.cumsum(task_pieces) - task_pieces
            n_pieces = 1 + int(task_pieces.sum())
            # (a task's stretch is addressed by 32-bit byte offsets)
            if n_pieces >= 2 ** 31 - 1 or int(task_pieces.max()) * 256 >= 2 ** 32:
                for k in ("tasks", "task_rows", "task_pairs", "task_out", "trk_meta"):
                    self.t[k] = None        # (the plan-less kernel takes over)
                return
            self.t["task_base"] = torch.from_numpy(base.astype(np.int32)).to(dev)
            # (+ a margin behind the last piece: a stager lane whose row has no
            # piece in a chunk loads from behind the chunk's stretch)
            self.t["frames"] = torch.empty(((n_pieces + 64) * 8, 4), dtype=torch.float64,
                                           device=dev)
            _lib.check(lib.taoamd_track_stream(
                self.n_tasks, _ptr(self.t["tasks"]), _ptr(self.t["task_rows"]),
                _ptr(self.t["trk_meta"]), _ptr(self.t["task_base"]), _ptr(padded),
                _ptr(self.t["frames"]), _stream()), "taoamd_track_stream")
            del padded
        # integer boxes: frame sums are exact in any order, nothing to guard
        # (products < 2^40, tracks of at most 2^12 frames: every sum < 2^53)
        longest = int((meta[:, 1] - meta[:, 0]).max()) + 1 if len(meta) else 0
        self.exact_terms = not bool(inexact.item()) and longest <= 4096

    def guard_active(self):
        """Whether a pass runs the frame-order guard: the count-based
        imagenetvid IoU is exact in any order, and so is the 3D IoU of integer
        boxes (exact_terms) -- but NOT the average IoU of integer boxes, whose
        per-frame ratios are rounded divisions.  One-frame tracks have a
        single term in every mode."""
        return (self.kind == "tao" and self.n_iou > 0 and self.iou_mode != 2
                and not (self.exact_terms and self.iou_mode == 0)
                and not self.single_frame and self.device.type == "cuda")

    def _plan_guard(self, flat, guard):
        """Tables of the frame-order guard (stage_iou_guard)."""
        if not self.guard_active():
            return
        lens = [int(np.diff(np.asarray(flat[s + "_frame_off"])).max())
                if len(flat[s + "_frame_off"]) > 1 else 0 for s in ("dt", "gt")]
        self.max_frames = lens
        # reordering bound of taoamd_track_iou_near (include/tao_amodal_hip.h)
        self.near_ulp = int(min(8 * (lens[0] + lens[1]) + 8, 2 ** 31 - 1))
        tl_id = np.ascontiguousarray(flat.tl_image_id, dtype=np.int64)
        ids_ok = len(tl_id) == 0 or (int(tl_id.min()) >= 0
                                     and int(tl_id.max()) < 2 ** 61 - 1)
        if guard == "device" and ids_ok:
            self.guard_on_device = True
            dev = self.device
            self.t["cell_unit"] = torch.from_numpy(
                np.ascontiguousarray(flat.cell_unit, dtype=np.int32)).to(dev)
            self.t["tl_vid_start"] = torch.from_numpy(
                np.ascontiguousarray(flat.tl_vid_start, dtype=np.int64)).to(dev)
            self.t["tl_image_id"] = torch.from_numpy(tl_id).to(dev)
        else:
            self.guard_flat = flat

    def input_bytes(self):
        return sum(v.numel() * v.element_size() for v in self.t.values()
                   if v is not None)


class Workspace:
    """Output and scratch buffers of one evaluator pass, allocated once."""

    def __init__(self, dp, detail=False, dt_rng_table=False, keep_order=False):
        lib = _lib.load()
        dev = dp.device
        u8 = torch.uint8

        def buf(nbytes):
            return torch.empty(max(int(nbytes), 256), dtype=u8, device=dev)
        self.gt_rng = torch.empty(max(dp.n_gt, 1), dtype=torch.int32, device=dev)
        # (image level: the match derives a detection's range mask from its
        # flags -- no table unless the caller wants to look at it)
        self.dt_rng = None if dp.kind == "lvis" and not (detail or dt_rng_table) \
            else torch.empty(max(dp.n_dt, 1), dtype=torch.int32, device=dev)
        self.num_gt = torch.empty((dp.n_cat, dp.n_rng), dtype=torch.int32,
                                  device=dev)
        # dst[i] = sorted place of detection i is what the pass needs (the match
        # writes a detection's row there); order[] = its inverse is derived on
        # demand (the `order` property) unless a caller wants the sort to store
        # it: 86 MB of stores less per pass at 21 M detections
        self.order_buf = torch.empty(max(dp.n_dt, 1), dtype=torch.int32, device=dev) \
            if (detail or keep_order or _os0.environ.get("TAOAMD_SORT_ASIDE", "0") != "0") else None
        self._n_dt = dp.n_dt
        self.dst = torch.empty(max(dp.n_dt, 1), dtype=torch.int32, device=dev)
        self.sort_bytes = max(lib.taoamd_sort_workspace(dp.n_dt),
                              lib.taoamd_sort_segments_workspace(dp.n_dt),
                              lib.taoamd_sort_sampled_workspace(
                                  dp.n_dt, dp.ss_sizes[3], int(dp.ss_merge)))
        self.sort_ws = buf(self.sort_bytes)
        # rows of (matched, ignored) pairs, one pair per 64 combos: the match
        # stores a pair with one 16-byte store and the sweep loads it with one
        # load (the C ABI sees two tables, the second one word behind the first)
        self.rows = torch.empty((max(dp.n_dt, 1), dp.n_words, 2),
                                dtype=torch.int64, device=dev)
        self.matched, self.ignored = self.rows[..., 0], self.rows[..., 1]
        self.acc_bytes = lib.taoamd_accumulate_workspace(dp.n_dt, dp.n_cat,
                                                         dp.n_rng)
        self.acc_ws = buf(self.acc_bytes)
        # the chunk table of the sweep depends on cat_off alone: built once here,
        # one launch less on the chain of every pass
        self.acc_prepared = None          # (max_segment hint, plan kind) it was built for
        self.sweep_recovered = 0          # passes swept again after a look-back gave up
        if torch.device(dev).type == "cuda":
            prepare_sweep(dp, self)
        self.precision = torch.empty((N_THR, N_REC, dp.n_cat, dp.n_rng),
                                     dtype=torch.float64, device=dev)
        self.recall = torch.empty((N_THR, dp.n_cat, dp.n_rng),
                                  dtype=torch.float64, device=dev)
        self.iou = None
        self.pair_frames = None
        if dp.kind == "tao":
            self.iou = torch.empty(max(dp.n_iou, 1), dtype=torch.float64,
                                   device=dev)
            self.pair_frames = torch.zeros(1, dtype=torch.int64, device=dev)
            self.near_count = torch.zeros(1, dtype=torch.int32, device=dev)
            self.guarded_pairs = 0
            # every pair can be listed: the list never overflows, so no pass
            # has to look at the count on the host
            self.near_cap = int(min(max(dp.n_iou, 1), 2 ** 31 - 1)) \
                if dp.guard_active() else 1
            self.near_list = torch.empty(self.near_cap, dtype=torch.int64, device=dev)
            if dp.guard_on_device:
                self.guard_table = int(lib.taoamd_track_iou_setorder_table(
                    *dp.max_frames))
                workers = GUARD_SCRATCH_BYTES // (12 * self.guard_table)
                workers = int(min(max(workers // 64 * 64, 64), 4096))
                self.guard_slots = workers * 3 * self.guard_table
                self.guard_scratch = torch.empty(self.guard_slots, dtype=torch.int32,
                                                 device=dev)
                self.guard_status = torch.zeros(1, dtype=torch.int32, device=dev)
        elif dp.mask_iou:
            self.iou = torch.empty(max(dp.n_iou, 1), dtype=torch.float64,
                                   device=dev)
            self.rle_bytes = lib.taoamd_rle_iou_workspace(
                dp.n_dt, dp.rle_total["dt"], dp.n_gt, dp.rle_total["gt"])
            self.rle_ws = buf(self.rle_bytes)
        self.match_gt = None
        self.ious_out = None
        if detail:
            self.match_gt = torch.empty((max(dp.n_dt, 1), dp.n_rng * N_THR),
                                        dtype=torch.int32, device=dev)
            if dp.kind == "lvis":
                self.ious_out = torch.empty(max(dp.n_iou, 1),
                                            dtype=torch.float64, device=dev)


def _order_of(ws):
    """order[p] = detection at sorted place p: stored by the sort when the
    workspace asked for it, else the inverse of dst[] (of the last pass)."""
    if ws.order_buf is not None:
        return ws.order_buf
    n = ws._n_dt
    out = torch.empty(max(n, 1), dtype=torch.int32, device=ws.dst.device)
    if n:
        out[ws.dst[:n].long()] = torch.arange(n, dtype=torch.int32, device=ws.dst.device)
    return out


Workspace.order = property(_order_of)


def stage_ranges(dp, ws):
    lib, t, s = _lib.load(), dp.t, _stream()
    if dp.kind == "lvis":
        _lib.check(lib.taoamd_lvis_ranges(
            dp.n_gt, _ptr(t["gt_vis"]), _ptr(t["gt_flags"]), _ptr(t["gt_cat"]),
            _ptr(t["gt_cat_off"]) if dp.grouped else None,
            dp.n_dt, _ptr(t["dt_flags"]), dp.n_cat, _ptr(ws.gt_rng),
            _ptr(ws.dt_rng), _ptr(ws.num_gt), s), "taoamd_lvis_ranges")
    else:
        _lib.check(lib.taoamd_tao_ranges(
            dp.n_gt, _ptr(t["gt_area"]), _ptr(t["gt_len"]), _ptr(t["gt_nhp"]),
            _ptr(t["gt_flags"]), _ptr(t["gt_cat"]),
            _ptr(t["gt_cat_off"]) if dp.grouped else None, dp.n_dt,
            _ptr(t["dt_area"]), _ptr(t["dt_len"]), _ptr(t["dt_flags"]),
            dp.n_cat, _ptr(ws.gt_rng), _ptr(ws.dt_rng), _ptr(ws.num_gt), s),
            "taoamd_tao_ranges")


import os as _os0
# Which segment sort a pass takes.  The sample sort (taoamd_sort_sampled: one
# scatter pass + one register sort per bucket) wins where categories span many
# LDS tiles -- 2000 videos, 21 M detections: 0.62 against 0.79 ms -- but is four
# dependent launches; below SORT_SAMPLED_MIN detections the tile sort + one
# rank-merge pass (taoamd_sort_segments) is the shorter chain.  (Until the
# bucket sort's exchanges became DPP moves the threshold was 6 M -- Config 2,
# 2.1 M detections: 107 against 130 us; since then the sample sort takes 80 us
# there against 102, and 67 against 77 at the 3 M rows of the stress stand-in.)
# TAOAMD_SORT=sampled / segments forces one.
SORT_SAMPLED_MIN = 1_500_000
_SORT_FORCE = _os0.environ.get("TAOAMD_SORT", "")


def sort_is_sampled(dp):
    if _SORT_FORCE in ("sampled", "segments"):
        return _SORT_FORCE == "sampled"
    return dp.n_dt >= SORT_SAMPLED_MIN


def stage_sort(dp, ws, order_only=False):
    lib, t, s = _lib.load(), dp.t, _stream()
    if dp.grouped and dp.n_dt and sort_is_sampled(dp):
        nc, ns, nt, nb = dp.ss_sizes
        _lib.check(lib.taoamd_sort_sampled(
            dp.n_dt, dp.n_cat, _ptr(t["cat_off"]), _ptr(t["tile_off"]), dp.n_tiles,
            dp.max_segment, _ptr(t["dt_score"]), nc, _ptr(t["ss_chunks"]), ns,
            _ptr(t["ss_split"]), nt, _ptr(t["ss_stile"]), nb, _ptr(t["ss_bucket"]),
            _ptr(ws.order_buf), None if order_only else _ptr(ws.dst), _ptr(ws.sort_ws),
            ws.sort_bytes, s),
            "taoamd_sort_sampled")
        return
    if dp.grouped:
        _lib.check(lib.taoamd_sort_segments(
            dp.n_dt, dp.n_cat, _ptr(t["cat_off"]), _ptr(t["tile_off"]),
            dp.n_tiles, dp.max_segment, _ptr(t["dt_cat"]), _ptr(t["dt_score"]),
            _ptr(ws.order_buf), _ptr(ws.dst), _ptr(ws.sort_ws), ws.sort_bytes, s),
            "taoamd_sort_segments")
        return
    _lib.check(lib.taoamd_sort_by_cat_score(
        dp.n_dt, _ptr(t["dt_cat"]), _ptr(t["dt_score"]), _ptr(ws.order_buf),
        _ptr(ws.dst), _ptr(ws.sort_ws), ws.sort_bytes, s),
        "taoamd_sort_by_cat_score")


def stage_mask_iou(dp, ws):
    """IoU of the run-length masks of every cell (iou_type="segm")."""
    if not dp.mask_iou or dp.n_iou == 0:
        return
    lib, t, s = _lib.load(), dp.t, _stream()
    _lib.check(lib.taoamd_rle_iou(
        dp.n_cells, _ptr(t["cell_dt_off"]), _ptr(t["cell_gt_off"]),
        _ptr(t["cell_iou_off"]), dp.n_dt, dp.rle_total["dt"],
        _ptr(t["dt_rle_off"]), _ptr(t["dt_rle_runs"]), _ptr(t["dt_rle_hw"]),
        _ptr(t["dt_rle_bb"]), dp.n_gt, dp.rle_total["gt"],
        _ptr(t["gt_rle_off"]), _ptr(t["gt_rle_runs"]), _ptr(t["gt_rle_hw"]),
        _ptr(t["gt_rle_bb"]), _ptr(ws.iou), _ptr(ws.rle_ws), ws.rle_bytes, s),
        "taoamd_rle_iou")


def stage_track_iou(dp, ws):
    if dp.kind == "lvis":
        return stage_mask_iou(dp, ws)        # the image level's pre-match IoU
    if dp.kind != "tao" or dp.n_iou == 0:
        return
    lib, t, s = _lib.load(), dp.t, _stream()
    if dp.single_frame:
        _lib.check(lib.taoamd_track_iou_single(
            dp.n_dt, _ptr(t["dt_group"]), _ptr(t["cell_iou_off"]),
            _ptr(t["dt_frame_pos"]), _ptr(t["dt_frame_box"]), _ptr(t["gt_frame_pos"]),
            _ptr(t["gt_frame_box"]), dp.iou_mode, _ptr(ws.iou), _ptr(ws.pair_frames), s),
            "taoamd_track_iou_single")
        return
    if t["tasks"] is not None:
        # the pairs' common frames (pair_frames: the unit the throughput is
        # quoted in) are a constant of the problem: counted by the first pass
        # over a workspace, which later passes leave as it is -- the 3D IoU's
        # arithmetic needs no frame masks, the kernel is 5 % shorter without them
        counted = getattr(ws, "pairs_counted", False) and dp.iou_mode == 0
        _lib.check(lib.taoamd_track_iou_planned(
            dp.n_tasks, _ptr(t["tasks"]), _ptr(t["task_rows"]),
            _ptr(t["task_pairs"]), _ptr(t["task_out"]), _ptr(t["frames"]),
            _ptr(t["task_base"]), _ptr(t["trk_meta"]), dp.iou_mode, _ptr(ws.iou),
            None if counted else _ptr(ws.pair_frames), s), "taoamd_track_iou_planned")
        ws.pairs_counted = True
        return
    _lib.check(lib.taoamd_track_iou(
        dp.n_cells, _ptr(t["cell_dt_off"]), _ptr(t["cell_gt_off"]),
        _ptr(t["cell_iou_off"]), dp.n_iou, _ptr(t["dt_frame_off"]),
        _ptr(t["dt_frame_pos"]), _ptr(t["dt_frame_box"]),
        _ptr(t["gt_frame_off"]), _ptr(t["gt_frame_pos"]),
        _ptr(t["gt_frame_box"]), dp.iou_mode, _ptr(ws.iou),
        _ptr(ws.pair_frames), s), "taoamd_track_iou")


GUARD_SCRATCH_BYTES = 64 << 20    # set tables of the device-side guard's threads


def stage_iou_guard(dp, ws):
    """The frame-order guard, all on the device: list the track pairs whose
    IoU a reordering of the frame sums could move across a comparison of the
    match (taoamd_track_iou_near, within dp.near_ulp of a threshold or of a
    rival), then recompute exactly those in the reference's CPython set order
    (taoamd_track_iou_setorder) and patch the IoU matrix -- asynchronous, no
    host round trip.  Nothing to do for integer boxes under the 3D IoU (exact
    sums) and for the count-based imagenetvid IoU."""
    if not dp.guard_active():
        return
    lib, t, s = _lib.load(), dp.t, _stream()
    _lib.check(lib.taoamd_track_iou_near(
        dp.n_cells, _ptr(t["cell_gt_off"]), _ptr(t["cell_iou_off"]), dp.n_iou,
        _ptr(ws.iou), dp.near_ulp, ws.near_cap, _ptr(ws.near_count),
        _ptr(ws.near_list), s), "taoamd_track_iou_near")
    if not dp.guard_on_device:
        return
    _lib.check(lib.taoamd_track_iou_setorder(
        dp.n_cells, _ptr(t["cell_dt_off"]), _ptr(t["cell_gt_off"]),
        _ptr(t["cell_iou_off"]), _ptr(t["cell_unit"]), _ptr(t["tl_vid_start"]),
        _ptr(t["tl_image_id"]), _ptr(t["dt_frame_off"]), _ptr(t["dt_frame_pos"]),
        _ptr(t["dt_frame_box"]), _ptr(t["gt_frame_off"]), _ptr(t["gt_frame_pos"]),
        _ptr(t["gt_frame_box"]), dp.iou_mode, _ptr(ws.near_count), ws.near_cap,
        _ptr(ws.near_list), _ptr(ws.iou), _ptr(ws.guard_scratch), ws.guard_slots,
        ws.guard_table, _ptr(ws.guard_status), s), "taoamd_track_iou_setorder")


def apply_iou_guard(dp, ws, flat=None):
    """Host half of the guard, for problems the device cannot recompute (an
    image id outside [0, 2^61 - 1): hash(id) != id) or that asked for it
    (DeviceProblem(guard="host"), the tests' cross-check): the listed pairs
    are recomputed with Python's own sets and patched into the IoU matrix
    before the match.  Returns the number of guarded pairs (synchronises)."""
    if not dp.guard_active() or dp.guard_on_device:
        return 0
    flat = flat if flat is not None else dp.guard_flat
    n = int(ws.near_count.item())
    if n == 0:
        return 0
    pairs = ws.near_list[:n].cpu().numpy()
    vals = set_order_iou(flat, pairs, dp.iou_mode)
    ws.iou[torch.from_numpy(pairs).to(dp.device)] = torch.from_numpy(vals).to(dp.device)
    return n


def _plan_key(dp):
    return (dp.acc_hint, int(_lib.load().taoamd_accumulate_plan_kind(
        dp.n_dt, dp.n_rng, dp.acc_hint)))


def prepare_sweep(dp, ws):
    """(Re)build the sweep's plan in the workspace (taoamd_accumulate_prepare):
    the table that cuts the categories into chunks / super-chunks under the
    current sweep mode."""
    with torch.cuda.device(dp.device):
        _lib.check(_lib.load().taoamd_accumulate_prepare(
            dp.n_dt, dp.n_cat, dp.n_rng, _ptr(dp.t["cat_off"]), dp.acc_hint,
            _ptr(ws.acc_ws), ws.acc_bytes, _stream()), "taoamd_accumulate_prepare")
    ws.acc_prepared = _plan_key(dp)


def sweep_flag(acc_ws, device):
    """The one-pass sweep's error flag of a workspace (taoamd_accumulate_error;
    synchronises with the current stream)."""
    import ctypes as C
    flag = C.c_int32(0)
    with torch.cuda.device(device):
        _lib.check(_lib.load().taoamd_accumulate_error(
            _ptr(acc_ws), _stream(), C.addressof(flag)), "taoamd_accumulate_error")
    return int(flag.value)


def sweep_ok(dp, ws):
    """Check the last pass on this workspace for the one-pass sweep's error flag
    (a look-back between workgroups gave up waiting; synchronises).  Such a
    pass is swept AGAIN with the chunked kernels -- its rows are still in the
    workspace -- so precision / recall are the right tables when this returns;
    the plan is rebuilt for the passes to come.  Returns True if it had to."""
    if dp.device.type != "cuda" or not sweep_flag(ws.acc_ws, dp.device):
        return False
    import logging
    logging.getLogger("tao_amodal_amd").warning(
        "the one-pass sweep's look-back between workgroups timed out (a busy or "
        "shared GPU?): this pass is swept again with the chunked kernels")
    lib, t = _lib.load(), dp.t
    with torch.cuda.device(dp.device):
        _lib.check(lib.taoamd_accumulate_chunked(
            dp.n_dt, dp.n_cat, dp.n_rng, _ptr(t["cat_off"]), _ptr(ws.matched),
            _ptr(ws.ignored), _ptr(ws.num_gt), dp.acc_hint, _ptr(ws.precision),
            _ptr(ws.recall), _ptr(ws.acc_ws), ws.acc_bytes, _stream()),
            "taoamd_accumulate_chunked")
    prepare_sweep(dp, ws)
    torch.cuda.synchronize(dp.device)
    ws.sweep_recovered += 1
    return True


def guarded_pairs(dp, ws, check_sweep=True):
    """Pairs the last pass listed and recomputed (synchronises).  Also the
    place where a pass is checked for the sweep's error flag (and swept again
    if it is set: sweep_ok; the multi-GPU plans check the workspace they swept
    on themselves, collectively: dist.ShardedEval.check)."""
    if check_sweep:
        sweep_ok(dp, ws)
    if not dp.guard_active():
        return 0
    if dp.guard_on_device and int(ws.guard_status.item()):
        raise _lib.TaoAmdError("frame-order guard: a track pair outgrew the set "
                               "tables sized for the longest tracks")
    n = int(ws.near_count.item())
    if n > max(dp.n_iou // 8, 1024):
        # every listed pair is recomputed serially by one thread (the CPython
        # set emulation): a large share of the pairs -- avg_iou on integer
        # boxes with many exact ties, duplicated tracks -- turns a 0.3 ms pass
        # into a much longer one without changing any result; say so
        import logging
        logging.getLogger("tao_amodal_amd").warning(
            "frame-order guard: %d of %d track pairs were recomputed in the "
            "reference's frame order (slow path)", n, dp.n_iou)
    return n


def set_order_iou(flat, pairs, mode=0):
    """3D IoU (mode 0) / average IoU (mode 1) of the listed pairs exactly as
    the reference computes them: {image id: box} maps in annotation order, the
    frames visited in the iteration order of
    ``set(gt.keys()) | set(dt.keys())`` (T/eval.py:73-117)."""
    ioff = np.asarray(flat.cell_iou_off)
    d_off, g_off = np.asarray(flat.cell_dt_off), np.asarray(flat.cell_gt_off)
    tl_id, tl_start = np.asarray(flat.tl_image_id), np.asarray(flat.tl_vid_start)
    unit = np.asarray(flat.cell_unit)
    fr = {}
    for side in ("dt", "gt"):
        fr[side] = (np.asarray(flat[side + "_frame_off"]),
                    np.asarray(flat[side + "_frame_pos"]),
                    np.asarray(flat[side + "_frame_box"]))
    cache = {}

    def track_map(side, t, v):
        key = (side, t)
        if key not in cache:
            off, pos, box = fr[side]
            a, b = int(off[t]), int(off[t + 1])
            ids = tl_id[tl_start[v] + pos[a:b]].tolist()
            cache[key] = dict(zip(ids, np.asarray(box[a:b]).tolist()))
        return cache[key]
    out = np.zeros(len(pairs))
    cells = np.searchsorted(ioff, pairs, "right") - 1
    for k, (p, c) in enumerate(zip(pairs.tolist(), cells.tolist())):
        G = int(g_off[c + 1] - g_off[c])
        local = p - int(ioff[c])
        v = int(unit[c])
        dmap = track_map("dt", int(d_off[c]) + local // G, v)
        gmap = track_map("gt", int(g_off[c]) + local % G, v)
        i = u = 0
        ious = []
        for im in set(gmap.keys()) | set(dmap.keys()):
            g, d = gmap.get(im), dmap.get(im)
            if d and g:
                w = max(min(d[0] + d[2], g[0] + g[2]) - max(d[0], g[0]), 0)
                h = max(min(d[1] + d[3], g[1] + g[3]) - max(d[1], g[1]), 0)
                i_ = w * h
                u_ = d[2] * d[3] + g[2] * g[3] - i_
                i += i_
                u += u_
                ious.append(i_ / u_ if u_ > 0 else 0)
            elif g:
                u += g[2] * g[3]
                ious.append(0)
            elif d:
                u += d[2] * d[3]
                ious.append(0)
        out[k] = (i / u if u > 0 else 0) if mode == 0 else float(np.mean(ious))
    return out


def stage_match(dp, ws, scatter=True, groups=None, singles=None):
    """`groups` / `singles`: (device table, first, count) -- a slice of a
    launch plan instead of the problem's whole plan (the phases of the
    multi-GPU exchange, dist.ShardedEval)."""
    if dp.n_dt == 0:        # nothing was detected: every cell is GT-only
        return
    lib, t, s = _lib.load(), dp.t, _stream()
    g_ptr, n_g = _ptr(t["groups"]), dp.n_groups
    s_ptr, n_s = _ptr(t["singles"]), dp.n_singles
    if groups is not None:
        g_ptr, n_g = groups[0].data_ptr() + 16 * groups[1], groups[2]
        s_ptr, n_s = singles[0].data_ptr() + 4 * singles[1], singles[2]
        if n_g == 0 and n_s == 0:
            return
    fused = dp.kind == "lvis" and not dp.mask_iou
    _lib.check(lib.taoamd_match(
        dp.n_cells, _ptr(t["cell_dt_off"]), _ptr(t["cell_gt_off"]),
        _ptr(t["cell_iou_off"]), dp.max_g,
        _ptr(t["dt_box"]) if fused else None,
        _ptr(t["gt_box"]) if fused else None,
        None if fused else _ptr(ws.iou), dp.n_rng, _ptr(ws.gt_rng),
        None if dp.kind == "lvis" else _ptr(ws.dt_rng),   # (image level: from the flags)
        _ptr(t["gt_flags"]), _ptr(t["dt_flags"]),
        _ptr(ws.dst) if scatter else None, 0, _ptr(ws.matched),
        _ptr(ws.ignored), _ptr(ws.match_gt), _ptr(ws.ious_out),
        _ptr(t["dt_group"]), _ptr(t["dt_meta"]), g_ptr, n_g,
        s_ptr, n_s, s), "taoamd_match")


def stage_accumulate_by_order(dp, ws):
    """The sweep over rows the match left in cell order (stage_match(scatter=
    False)): gathered through the sort's order[] by the first sweep."""
    lib, t, s = _lib.load(), dp.t, _stream()
    _lib.check(lib.taoamd_accumulate_by_order(
        dp.n_dt, dp.n_cat, dp.n_rng, _ptr(t["cat_off"]), _ptr(ws.order),
        _ptr(ws.matched), _ptr(ws.ignored), _ptr(ws.num_gt), dp.acc_hint,
        _ptr(ws.precision), _ptr(ws.recall), _ptr(ws.acc_ws), ws.acc_bytes, s),
        "taoamd_accumulate_by_order")


def stage_accumulate(dp, ws):
    lib, t, s = _lib.load(), dp.t, _stream()
    # (a plan built under another sweep mode or hint does not serve this pass)
    fn = lib.taoamd_accumulate_prepared if ws.acc_prepared == _plan_key(dp) \
        else lib.taoamd_accumulate
    _lib.check(fn(
        dp.n_dt, dp.n_cat, dp.n_rng, _ptr(t["cat_off"]), _ptr(ws.matched),
        _ptr(ws.ignored), _ptr(ws.num_gt), dp.acc_hint, _ptr(ws.precision),
        _ptr(ws.recall), _ptr(ws.acc_ws), ws.acc_bytes, s),
        "taoamd_accumulate")


def stage_track_iou_guarded(dp, ws):
    stage_track_iou(dp, ws)
    stage_iou_guard(dp, ws)


STAGES = (("ranges", stage_ranges), ("sort", stage_sort),
          ("track_iou", stage_track_iou_guarded), ("match", stage_match),
          ("accumulate", stage_accumulate))


def run(dp, ws):
    """Launch one evaluator pass on the current stream (asynchronous, unless
    the problem needs the host half of the frame-order guard)."""
    if _lib.TIMING:
        _lib.kernel_timing_label(dp.kind)
    for name, fn in STAGES:
        fn(dp, ws)
        if name == "track_iou" and dp.guard_flat is not None:
            ws.guarded_pairs = apply_iou_guard(dp, ws)


import os as _os
# Overlap: image-level sort beside ranges + match (see run_forked).  Off: measured
# slower on one MI355X (0.439 vs 0.421 ms/step at Config 2: the match takes 139
# instead of 84 us beside the sort -- the step is bound by the sum of its
# kernels, not by the chain); TAOAMD_SORT_ASIDE=1 switches it on for A/B timing
SORT_ASIDE = _os.environ.get("TAOAMD_SORT_ASIDE", "0") != "0"
# A/B: the 3D IoU (the longest kernel) alone ahead of the image level instead of beside it
TRACK_FIRST = _os.environ.get("TAOAMD_TRACK_FIRST", "0") != "0"
TRACK_AFTER_SORT = _os.environ.get("TAOAMD_TRACK_AFTER_SORT", "0") != "0"
# ... or when its splitter kernel is done (taoamd_sort_sampled_notify): the
# splitters then take 0.08 instead of 0.35 ms, but the 3D IoU runs beside the
# scatter and the bucket sort instead and slows those down: 1.54 against 1.52
# ms.  The step is bound by the sum of its kernels' work, not by the chain.
TRACK_AFTER_SPLIT = _os.environ.get("TAOAMD_TRACK_AFTER_SPLIT", "0") != "0"
# ... or when the image level's MATCH is done: the 3D IoU (bound by the LDS
# array) then runs beside the sweep (bound by VALU issue, no LDS to speak of) --
# the one pairing of the step's long kernels that does not share its bound
TRACK_AFTER_MATCH = _os.environ.get("TAOAMD_TRACK_AFTER_MATCH", "0") != "0"


class Overlap:
    """Runs evaluator passes concurrently on separate HIP streams.

    Every kernel of this path is latency / dependency bound (rocprofv3 SQ
    counters: 20-45 % active cycles), so independent work fills the idle
    issue slots: the image-level and the track-level pass share nothing, and
    inside a pass range masks, sort and 3D IoU only meet at the match kernel.
    Streams are forked from / joined to the caller's current stream with
    events, so the caller's barrier + synchronize bracketing stays valid."""

    def __init__(self, device, n=4):
        import ctypes as C
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(self.device) for _ in range(n)]
        # recorded behind the image level's splitter kernel: what the track
        # level waits for (run_pair)
        self.split_done = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().taoamd_event_create(C.byref(self.split_done)),
                       "taoamd_event_create")

    def __del__(self):
        try:
            if self.split_done:
                _lib.load().taoamd_event_destroy(self.split_done)
        except Exception:
            pass

    def _fork(self, k):
        s = self.streams[k]
        s.wait_stream(torch.cuda.current_stream(self.device))
        return s

    def run_pair(self, dpl, wsl, dpt, wst):
        # The image-level chain is the longer one (sort -> match -> six sweep
        # kernels): it stays on the caller's stream, so nothing on it waits
        # for a cross-stream event at the step's start or end (measured:
        # 0.479 -> 0.464 ms against forking both passes); only the track-level
        # pass is forked and joined.
        cur = torch.cuda.current_stream(self.device)
        s_aux_l, s_aux_t = self.streams[2], self.streams[3]   # forked by run_forked
        if TRACK_FIRST and dpt.kind == "tao":
            # A/B schedule: the track level's head (range masks, sort, 3D IoU)
            # ahead of everything, the rest of it beside the image level
            run_forked(dpt, wst, s_aux_t, head_only=True, no_match=True)
            st = self._fork(1)
            run_forked(dpl, wsl, s_aux_l, sort_aside=SORT_ASIDE)
            with torch.cuda.stream(st):
                _probed("match", stage_match, dpt, wst)
                stage_accumulate(dpt, wst)
            cur.wait_stream(st)
            return
        st = self._fork(1)
        if TRACK_AFTER_SPLIT and dpl.kind == "lvis" and sort_is_sampled(dpl) and dpl.grouped \
                and dpl.n_dt and not SORT_ASIDE:
            # The track level starts when the image level's SPLITTERS are done.
            # Those are 0.06 ms of latency at the head of the step's critical
            # chain, 1200 workgroups of 139 VGPRs; started together, the 3D
            # IoU's 11 k small workgroups took every wave slot that came free
            # and the splitters ran for 0.35 ms (kernel trace of round 4).
            lib = _lib.load()
            lib.taoamd_sort_sampled_notify(self.split_done)
            run_forked(dpl, wsl, s_aux_l)
            _lib.check(lib.taoamd_stream_wait_event(st.cuda_stream, self.split_done),
                       "taoamd_stream_wait_event")
            _lib.check(lib.taoamd_stream_wait_event(s_aux_t.cuda_stream, self.split_done),
                       "taoamd_stream_wait_event")
        elif TRACK_AFTER_MATCH:
            ev = torch.cuda.Event()
            run_forked(dpl, wsl, s_aux_l, sort_aside=SORT_ASIDE, match_done=ev)
            # (the image level's sweep is launched by run_forked right behind the
            # event: the track level's kernels queue beside it)
            st.wait_event(ev)
        elif TRACK_AFTER_SORT:
            # A/B schedule: the track level starts when the image level's sort is
            # done -- its 3D IoU then runs beside the match and the sweep instead
            # of holding the wave slots the sort's short kernels wait for
            ev = torch.cuda.Event()
            run_forked(dpl, wsl, s_aux_l, sort_aside=SORT_ASIDE, sort_done=ev)
            st.wait_event(ev)
        else:
            run_forked(dpl, wsl, s_aux_l, sort_aside=SORT_ASIDE)
        with torch.cuda.stream(st):
            run_forked(dpt, wst, s_aux_t)
        cur.wait_stream(st)


class StageProbe:
    """HIP events around single-kernel stages, recorded on the stream the
    kernel is launched on while the (overlapped) steps run: the dominant
    kernels' launch durations inside the timed region of bench.py."""

    def __init__(self):
        self.events = {}

    def wrap(self, name, fn, dp, ws):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        fn(dp, ws)
        b.record()
        self.events.setdefault(name, []).append((a, b))

    def mean_ms(self):
        """(call after a synchronize)"""
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v)
                for k, v in self.events.items()}


PROBE = None     # set to a StageProbe to time stage_track_iou / stage_match


def _probed(name, fn, dp, ws):
    if PROBE is None:
        fn(dp, ws)
    else:
        PROBE.wrap(dp.kind + ":" + name, fn, dp, ws)


def run_forked(dp, ws, aux, head_only=False, sort_aside=False, no_match=False,
               sort_done=None, match_done=None):
    """One evaluator pass on the current stream, its independent head stages
    on the stream `aux` (forked from and joined to the current stream):
    image level  ranges || sort -> match -> accumulate,
    track level  (ranges, sort) || 3D IoU -> match -> accumulate.
    `sort_aside` (image level): sort || (ranges -> match, rows left in cell
    order) -> accumulate gathering the rows through order[] -- the match does
    not wait for the sort and stores full wavefront runs."""
    cur = torch.cuda.current_stream(dp.device)
    aux.wait_stream(cur)
    if _lib.TIMING:
        _lib.kernel_timing_label(dp.kind)
    if sort_aside and dp.kind == "lvis" and not dp.mask_iou and not head_only \
            and dp.grouped and dp.n_dt:
        with torch.cuda.stream(aux):
            stage_sort(dp, ws, order_only=sort_is_sampled(dp) and ws.order_buf is not None)
        stage_ranges(dp, ws)
        _probed("match", lambda d, w: stage_match(d, w, scatter=False), dp, ws)
        cur.wait_stream(aux)
        stage_accumulate_by_order(dp, ws)
        return
    if dp.kind == "lvis":
        with torch.cuda.stream(aux):
            stage_ranges(dp, ws)
            stage_mask_iou(dp, ws)
        stage_sort(dp, ws)
        if sort_done is not None:
            sort_done.record(cur)
    else:
        with torch.cuda.stream(aux):
            stage_ranges(dp, ws)
            stage_sort(dp, ws)
        _probed("track_iou", stage_track_iou_guarded, dp, ws)
        if dp.guard_flat is not None:
            # host half of the guard (ids Python does not hash to themselves):
            # one synchronisation, the listed pairs patched before the match
            ws.guarded_pairs = apply_iou_guard(dp, ws)
    cur.wait_stream(aux)
    if no_match:
        return
    _probed("match", stage_match, dp, ws)
    if match_done is not None:
        match_done.record(cur)
    if not head_only:
        stage_accumulate(dp, ws)


def time_stages(dpl, wsl, dpt, wst, reps=10):
    """Average duration (ms) of every stage of both evaluators, measured with
    HIP events recorded on the stream the kernels are launched on."""
    out = {}
    for side, dp, ws in (("lvis", dpl, wsl), ("tao", dpt, wst)):
        acc = {name: 0.0 for name, _ in STAGES}
        for _ in range(reps):
            evs = []
            for name, fn in STAGES:
                a = torch.cuda.Event(enable_timing=True)
                b = torch.cuda.Event(enable_timing=True)
                a.record()
                fn(dp, ws)
                b.record()
                evs.append((name, a, b))
            torch.cuda.synchronize(dp.device)
            for name, a, b in evs:
                acc[name] += a.elapsed_time(b)
        out[side] = {k: round(v / reps, 4) for k, v in acc.items()}
    return out


def run_guarded(dp, ws, flat=None, upto=None, read_count=True, head=True):
    """One evaluator pass; returns the number of pairs the frame-order guard
    recomputed (synchronises at the end to read it, unless read_count=False:
    guarded_pairs() gives it later).  `head=False`: a further pass over the
    SAME detections under other evaluation constants (blocks of thresholds /
    ranges, evaluation/_core.py) -- the score order and the IoU matrix of the
    pass before it stand (neither depends on a threshold or a range); the
    range masks, the guard's near list (it is taken against the thresholds;
    pairs it patched earlier keep the reference-order value) and the match
    are what is repeated."""
    if _lib.TIMING:
        _lib.kernel_timing_label(dp.kind)
    stage_ranges(dp, ws)
    if head:
        stage_sort(dp, ws)
        stage_track_iou(dp, ws)
    stage_iou_guard(dp, ws)
    apply_iou_guard(dp, ws, flat)
    stage_match(dp, ws)
    if upto != "match":
        stage_accumulate(dp, ws)
    if not read_count:
        return None
    ws.guarded_pairs = guarded_pairs(dp, ws)
    return ws.guarded_pairs


def evaluate_flat(flat, device="cuda", detail=False, iou_3d_type="3d_iou",
                  guard="device"):
    """Upload, run, download.  Returns a dict of numpy arrays shaped like the
    C oracle's outputs (tests compare the two field by field)."""
    dp = DeviceProblem(flat, device, iou_3d_type, guard=guard)
    ws = Workspace(dp, detail=detail, dt_rng_table=True)
    guarded = run_guarded(dp, ws, flat)
    torch.cuda.synchronize(dp.device)
    n = dp.n_dt
    out = {
        "precision": ws.precision.cpu().numpy(),
        "recall": ws.recall.cpu().numpy(),
        "num_gt": ws.num_gt.cpu().numpy(),
        "gt_rng": ws.gt_rng[:dp.n_gt].cpu().numpy().view(np.uint32),
        "dt_rng": ws.dt_rng[:n].cpu().numpy().view(np.uint32),
        "order": ws.order[:n].cpu().numpy().astype(np.int64),
        "dst": ws.dst[:n].cpu().numpy().astype(np.int64),
        # rows are in sorted order on the device; give them back per detection
        "matched_sorted": ws.matched[:n].cpu().numpy().view(np.uint64),
        "ignored_sorted": ws.ignored[:n].cpu().numpy().view(np.uint64),
    }
    out["matched"] = out["matched_sorted"][out["dst"]] if n else out["matched_sorted"]
    out["ignored"] = out["ignored_sorted"][out["dst"]] if n else out["ignored_sorted"]
    if dp.kind == "tao":
        out["iou"] = ws.iou[:dp.n_iou].cpu().numpy()
        out["pairs"] = int(ws.pair_frames.item())
        out["near_threshold_pairs"] = guarded
    if detail:
        out["match_gt"] = ws.match_gt[:n].cpu().numpy()
        if dp.kind == "lvis" and not dp.mask_iou:
            out["iou"] = ws.ious_out[:dp.n_iou].cpu().numpy()
    if dp.mask_iou:
        out["iou"] = ws.iou[:dp.n_iou].cpu().numpy()
    return out
