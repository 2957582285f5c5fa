"""Multi-GPU evaluation: one process per GPU, units (images / videos) sharded
across ranks, RCCL (``torch.distributed`` backend "nccl") over xGMI.

Why an exchange is needed at all: IoU + greedy match is
independent per cell, but AP needs every category's detections in ONE global
score order (reference lvis_amodal/eval.py:353-361).  A histogram all-reduce
is not enough for exact AP; the per-detection TP/ignore words have to meet.

Per evaluator pass and rank:

  1. range masks; ``all_reduce(sum)`` of num_gt[K, n_rng]        (tiny)
  2. [3D track IoU]; match kernel writes each detection's words straight
     into its slot of the send buffer (rows grouped by owner rank, local
     order kept) -- no pack pass
  3. ONE ``all_to_all_single`` of the records
         [score | category | matched words | ignored words]   (int64 cols)
     Category k is owned by rank k // ceil(K / world): contiguous blocks.
  4. owner: stable (category, -score) sort of what it received.  Ranks hold
     ascending, disjoint unit ranges and all_to_all delivers sources in rank
     order, so "received order" == the reference's concatenation order and
     the stable sort reproduces its tie-breaking exactly.
  5. owner: gather rows into sorted order, sweep its categories
     (taoamd_accumulate_compact) into the category-major tables
  6. ONE in-place ``all_gather_into_tensor`` of the tables (a rank's share is
     one contiguous block), then every rank transposes them into the
     reference layout (taoamd_finalize).

Volumes at Config 2 per rank: step 3 ~ 32 B x 2.1 M records, step 6 ~ 250 MB
/ world per rank -- far below the 7 x ~153 GB/s xGMI links, so the design
minimises the NUMBER of collectives (2 + one tiny all-reduce per evaluator).

The collective plumbing is backend-agnostic: ``tests/test_dist_gloo.py`` runs
this very module with world_size 2 on CPU tensors over gloo, with the oracle
standing in for the kernels.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib

N_THR, N_REC = _lib.N_THR, _lib.N_REC


def _ptr(t):
    return None if t is None else t.data_ptr()


class HipBackend:
    """The kernels of the C ABI on the current HIP stream."""

    def __init__(self):
        from . import engine
        self.engine = engine
        self.lib = _lib.load()

    def _s(self):
        return torch.cuda.current_stream().cuda_stream

    def ranges(self, dp, ws):
        self.engine.stage_ranges(dp, ws)

    def track_iou(self, dp, ws):
        self.engine.stage_track_iou(dp, ws)

    def match_into(self, dp, ws, dst, records, width):
        if dp.n_dt == 0:
            return
        t, lib = dp.t, self.lib
        fused = dp.kind == "lvis"
        base = records.data_ptr()
        _lib.check(lib.taoamd_match(
            dp.n_cells, _ptr(t["cell_dt_off"]), _ptr(t["cell_gt_off"]),
            _ptr(t["cell_iou_off"]), dp.max_g,
            _ptr(t["dt_box"]) if fused else None,
            _ptr(t["gt_box"]) if fused else None,
            None if fused else _ptr(ws.iou), dp.n_rng, _ptr(ws.gt_rng),
            _ptr(ws.dt_rng), _ptr(t["gt_flags"]), _ptr(t["dt_flags"]),
            _ptr(dst), width, base + 16, base + 16 + 8 * dp.n_words, None, None,
            _ptr(t["dt_cell"]), _ptr(t["groups"]), dp.n_groups,
            _ptr(t["singles"]), dp.n_singles, self._s()), "taoamd_match")

    def sort(self, n, cat, score, order, ws_buf, ws_bytes):
        _lib.check(self.lib.taoamd_sort_by_cat_score(
            n, _ptr(cat), _ptr(score), _ptr(order), None, _ptr(ws_buf),
            ws_bytes, self._s()), "taoamd_sort_by_cat_score")

    def gather_rows(self, n, n_words, records, width, order, matched, ignored):
        base = records.data_ptr()
        _lib.check(self.lib.taoamd_gather_rows(
            n, n_words, base + 16, base + 16 + 8 * n_words, width, _ptr(order),
            _ptr(matched), _ptr(ignored), self._s()), "taoamd_gather_rows")

    def accumulate_compact(self, n, n_cat, n_rng, cat_off, matched, ignored,
                           num_gt, k0, k1, val, rec, ws_buf, ws_bytes):
        _lib.check(self.lib.taoamd_accumulate_compact(
            n, n_cat, n_rng, _ptr(cat_off), _ptr(matched), _ptr(ignored),
            _ptr(num_gt), k0, k1, _ptr(val), _ptr(rec), _ptr(ws_buf), ws_bytes,
            self._s()), "taoamd_accumulate_compact")

    def finalize(self, n_cat, n_rng, num_gt, val, rec, precision, recall):
        _lib.check(self.lib.taoamd_finalize(
            n_cat, n_rng, _ptr(num_gt), _ptr(val), _ptr(rec), _ptr(precision),
            _ptr(recall), self._s()), "taoamd_finalize")


class ShardedEval:
    """One evaluator (LVIS or TAO side) of one rank."""

    def __init__(self, dp, ws, rank, world, backend, group=None):
        self.dp, self.ws, self.rank, self.world = dp, ws, rank, world
        self.be, self.group = backend, group
        dev = dp.device
        K, nw = dp.n_cat, dp.n_words
        self.W = 2 + 2 * nw                       # int64 columns of a record
        self.Kb = (K + world - 1) // world        # categories per owner block
        cat = dp.t["dt_cat"].to(torch.int64)
        owner = torch.div(cat, self.Kb, rounding_mode="floor")
        # stable partition by owner: slot of every local detection
        perm = torch.argsort(owner, stable=True)
        slot = torch.empty_like(perm)
        slot[perm] = torch.arange(dp.n_dt, device=dev)
        self.dst = slot.to(torch.int32)
        self.send_counts = torch.bincount(owner, minlength=world)[:world].cpu().tolist()
        counts = torch.tensor(self.send_counts, dtype=torch.int64, device=dev)
        rc = torch.empty_like(counts)
        dist.all_to_all_single(rc, counts, group=group)
        self.recv_counts = rc.cpu().tolist()
        self.n_recv = int(sum(self.recv_counts))
        self.send = torch.zeros((max(dp.n_dt, 1), self.W), dtype=torch.int64,
                                device=dev)
        self.recv = torch.zeros((max(self.n_recv, 1), self.W), dtype=torch.int64,
                                device=dev)
        # static columns: score bits and category
        if dp.n_dt:
            self.send[slot, 0] = dp.t["dt_score"].view(torch.int64)
            self.send[slot, 1] = cat
        self._exchange()
        rcat = self.recv[:self.n_recv, 1]
        cat_off = torch.zeros(K + 1, dtype=torch.int64, device=dev)
        if self.n_recv:
            cat_off[1:] = torch.cumsum(torch.bincount(rcat, minlength=K)[:K], 0)
        self.cat_off = cat_off.to(torch.int32)
        self.k0 = min(rank * self.Kb, K)
        self.k1 = min((rank + 1) * self.Kb, K)
        lib = _lib.load()
        n = max(self.n_recv, 1)
        self.rcat = torch.empty(n, dtype=torch.int32, device=dev)
        self.rscore = torch.empty(n, dtype=torch.float64, device=dev)
        self.order = torch.empty(n, dtype=torch.int32, device=dev)
        self.sort_bytes = lib.taoamd_sort_workspace(self.n_recv)
        self.sort_ws = torch.empty(max(self.sort_bytes, 256), dtype=torch.uint8,
                                   device=dev)
        self.matched = torch.empty((n, nw), dtype=torch.int64, device=dev)
        self.ignored = torch.empty((n, nw), dtype=torch.int64, device=dev)
        self.acc_bytes = lib.taoamd_accumulate_workspace(self.n_recv, K, dp.n_rng)
        self.acc_ws = torch.empty(max(self.acc_bytes, 256), dtype=torch.uint8,
                                  device=dev)
        kpad = self.Kb * world
        self.val = torch.zeros((kpad, dp.n_rng, N_THR, N_REC),
                               dtype=torch.float64, device=dev)
        self.rec = torch.zeros((kpad, dp.n_rng, N_THR), dtype=torch.float64,
                               device=dev)
        self.num_gt = torch.zeros((K, dp.n_rng), dtype=torch.int32, device=dev)
        self.precision = torch.empty((N_THR, N_REC, K, dp.n_rng),
                                     dtype=torch.float64, device=dev)
        self.recall = torch.empty((N_THR, K, dp.n_rng), dtype=torch.float64,
                                  device=dev)

    def _exchange(self):
        dist.all_to_all_single(
            self.recv[:self.n_recv], self.send[:self.dp.n_dt],
            output_split_sizes=self.recv_counts,
            input_split_sizes=self.send_counts, group=self.group)

    def step(self):
        dp, ws, be = self.dp, self.ws, self.be
        be.ranges(dp, ws)
        self.num_gt.copy_(ws.num_gt)
        dist.all_reduce(self.num_gt, group=self.group)
        be.track_iou(dp, ws)
        be.match_into(dp, ws, self.dst, self.send, self.W)
        self._exchange()
        n = self.n_recv
        if n:
            self.rcat[:n].copy_(self.recv[:n, 1])
            self.rscore[:n].copy_(self.recv[:n, 0].view(torch.float64))
        be.sort(n, self.rcat, self.rscore, self.order, self.sort_ws,
                self.sort_bytes)
        be.gather_rows(n, dp.n_words, self.recv, self.W, self.order,
                       self.matched, self.ignored)
        be.accumulate_compact(n, dp.n_cat, dp.n_rng, self.cat_off, self.matched,
                              self.ignored, self.num_gt, self.k0, self.k1,
                              self.val, self.rec, self.acc_ws, self.acc_bytes)
        lo, hi = self.rank * self.Kb, (self.rank + 1) * self.Kb
        dist.all_gather_into_tensor(self.val, self.val[lo:hi], group=self.group)
        dist.all_gather_into_tensor(self.rec, self.rec[lo:hi], group=self.group)
        be.finalize(dp.n_cat, dp.n_rng, self.num_gt, self.val, self.rec,
                    self.precision, self.recall)


class ExchangePlan:
    """Both evaluators of one rank (what bench.py steps)."""

    def __init__(self, dpl, dpt, rank, world, device, backend=None, group=None):
        from . import engine
        backend = backend or HipBackend()
        self.lvis = ShardedEval(dpl, engine.Workspace(dpl), rank, world, backend,
                                group)
        self.tao = ShardedEval(dpt, engine.Workspace(dpt), rank, world, backend,
                               group)

    def pair_frames(self):
        return int(self.tao.ws.pair_frames.item())


def step(plan):
    plan.lvis.step()
    plan.tao.step()


# --------------------------------------------------------------------------
# sharding one flattened problem (class API on several GPUs)
# --------------------------------------------------------------------------
def shard_bounds(flat, world):
    """Contiguous unit ranges (images or videos, in sorted order) with roughly
    equal numbers of box pairs.  Returns cell boundaries, len world + 1."""
    d = np.diff(flat.cell_dt_off).astype(np.int64)
    g = np.diff(flat.cell_gt_off).astype(np.int64)
    cost = np.cumsum(d * g + d + g)
    total = cost[-1] if len(cost) else 0
    bounds = [0]
    for r in range(1, world):
        c = int(np.searchsorted(cost, total * r / world))
        # move to the next unit boundary
        while 0 < c < flat.n_cells and flat.cell_unit[c] == flat.cell_unit[c - 1]:
            c += 1
        bounds.append(max(min(c, flat.n_cells), bounds[-1]))
    bounds.append(flat.n_cells)
    return bounds


def shard_flat(flat, c0, c1):
    """The sub-problem made of cells [c0, c1)."""
    from .flatten import Flat
    f = Flat()
    d0, d1 = int(flat.cell_dt_off[c0]), int(flat.cell_dt_off[c1])
    g0, g1 = int(flat.cell_gt_off[c0]), int(flat.cell_gt_off[c1])
    per_dt = ["dt_score", "dt_flags", "dt_id", "dt_cat", "dt_cell"]
    per_gt = ["gt_flags", "gt_id", "gt_cat", "gt_cell"]
    if flat.kind == "lvis":
        per_dt += ["dt_box"]
        per_gt += ["gt_box", "gt_vis"]
    else:
        per_dt += ["dt_area", "dt_len"]
        per_gt += ["gt_area", "gt_len", "gt_nhp"]
    for k, v in flat.items():
        if k in per_dt:
            f[k] = v[d0:d1]
        elif k in per_gt:
            f[k] = v[g0:g1]
        elif k in ("cell_unit", "cell_cat", "cell_span"):
            f[k] = v[c0:c1]
        elif k in ("cell_dt_off", "cell_gt_off", "cell_iou_off"):
            f[k] = (v[c0:c1 + 1] - v[c0]).astype(v.dtype)
        elif not k.startswith(("dt_frame", "gt_frame")):
            f[k] = v
    f.dt_cell = f.dt_cell - c0
    f.gt_cell = f.gt_cell - c0
    f.n_cells = c1 - c0
    if flat.kind == "tao":
        for side, a, b in (("dt", d0, d1), ("gt", g0, g1)):
            off = flat[side + "_frame_off"]
            lo, hi = int(off[a]), int(off[b])
            f[side + "_frame_off"] = (off[a:b + 1] - lo).astype(off.dtype)
            f[side + "_frame_pos"] = flat[side + "_frame_pos"][lo:hi]
            f[side + "_frame_box"] = flat[side + "_frame_box"][lo:hi]
        f.n_pairs = int(f.cell_iou_off[-1])
    else:
        f.n_pairs = int(np.sum(np.diff(f.cell_dt_off).astype(np.int64)
                               * np.diff(f.cell_gt_off)))
    return f
