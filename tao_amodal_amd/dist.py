"""Multi-GPU evaluation: one process per GPU, RCCL (``torch.distributed``
backend "nccl") over xGMI.  Two partitions of the path are implemented:

* BY CATEGORY (``CategoryPlan``, default of bench.py): the match and the AP
  sweep are independent per category and the cell tables are category-major,
  so a rank evaluates a contiguous category block with no record exchange at
  all; the only collectives are the in-place all-gathers that assemble the
  category-major result tables.  This is the natural mode when one
  prediction file is evaluated on several GPUs.
* BY UNIT (``ExchangePlan``): every rank holds the detections of its own
  images / videos (e.g. produced by data-parallel inference) and the records
  meet at the category owners -- described next.

Units (images / videos) sharded across ranks:

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
        fused = dp.kind == "lvis" and not dp.mask_iou
        base = records.data_ptr()
        _lib.check(lib.taoamd_match(
            dp.n_cells, _ptr(t["cell_dt_off"]), _ptr(t["cell_gt_off"]),
            _ptr(t["cell_iou_off"]), dp.max_g,
            _ptr(t["dt_box"]) if fused else None,
            _ptr(t["gt_box"]) if fused else None,
            None if fused else _ptr(ws.iou), dp.n_rng, _ptr(ws.gt_rng),
            _ptr(ws.dt_rng), _ptr(t["gt_flags"]), _ptr(t["dt_flags"]),
            _ptr(dst), width, base + 16, base + 16 + 8 * dp.n_words, None, None,
            _ptr(t["dt_group"]), _ptr(t["groups"]), dp.n_groups,
            _ptr(t["singles"]), dp.n_singles, self._s()), "taoamd_match")

    def sort_local(self, dp, ws):
        self.engine.stage_sort(dp, ws)

    def local_head(self, dp, ws, aux):
        """ranges, sort, [3D IoU], match with the independent stages on `aux`."""
        self.engine.run_forked(dp, ws, aux, head_only=True)

    def match_local(self, dp, ws):
        self.engine.stage_match(dp, ws)

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
                           num_gt, k0, k1, val, rec, ws_buf, ws_bytes,
                           max_segment=0):
        _lib.check(self.lib.taoamd_accumulate_compact(
            n, n_cat, n_rng, _ptr(cat_off), _ptr(matched), _ptr(ignored),
            _ptr(num_gt), k0, k1, max_segment, _ptr(val), _ptr(rec),
            _ptr(ws_buf), ws_bytes, self._s()), "taoamd_accumulate_compact")

    def finalize(self, n_cat, n_rng, num_gt, val, rec, precision, recall):
        _lib.check(self.lib.taoamd_finalize(
            n_cat, n_rng, _ptr(num_gt), _ptr(val), _ptr(rec), _ptr(precision),
            _ptr(recall), self._s()), "taoamd_finalize")

    # ---- result exchange of the category-partitioned mode
    def exchange_chunk_bytes(self, block_cats, n_rng, capacity):
        return int(self.lib.taoamd_exchange_chunk_bytes(block_cats, n_rng, capacity))

    def exchange_workspace(self, block_cats, n_rng, world):
        return int(self.lib.taoamd_exchange_workspace(block_cats, n_rng, world))

    def exchange_sizes(self, block_cats, n_rng, world, num_gt, totals, xws):
        _lib.check(self.lib.taoamd_exchange_sizes(
            block_cats, n_rng, world, _ptr(num_gt), _ptr(totals), _ptr(xws),
            xws.numel(), self._s()), "taoamd_exchange_sizes")

    def exchange_pack(self, n_cat, n_rng, block_cats, world, rank, num_gt, val,
                      rec, chunk, capacity, overflow, xws):
        _lib.check(self.lib.taoamd_exchange_pack(
            n_cat, n_rng, block_cats, world, rank, _ptr(num_gt), _ptr(val),
            _ptr(rec), _ptr(chunk), capacity, _ptr(overflow), _ptr(xws),
            xws.numel(), self._s()), "taoamd_exchange_pack")

    def exchange_unpack(self, n_cat, n_rng, block_cats, world, chunks, capacity,
                        num_gt, precision, recall, overflow, xws):
        _lib.check(self.lib.taoamd_exchange_unpack(
            n_cat, n_rng, block_cats, world, _ptr(chunks), capacity,
            _ptr(num_gt), _ptr(precision), _ptr(recall), _ptr(overflow),
            _ptr(xws), xws.numel(), self._s()), "taoamd_exchange_unpack")


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


def category_block(n_cat, rank, world):
    """Contiguous, equally sized category blocks: rank r owns [k0, k1)."""
    kb = (n_cat + world - 1) // world
    return min(rank * kb, n_cat), min((rank + 1) * kb, n_cat), kb


def shard_by_category(flat, k0, k1):
    """Cells of the categories [k0, k1) -- one contiguous slice of the
    category-major cell table."""
    c0 = int(np.searchsorted(flat.cell_cat, k0, "left"))
    c1 = int(np.searchsorted(flat.cell_cat, k1, "left"))
    return shard_flat(flat, c0, c1)


class CategoryShardedEval:
    """One evaluator of one rank when the problem is partitioned BY CATEGORY.

    A category's cells, detections and ground truth are one contiguous slice
    of the category-major cell tables (shard_by_category), and both the greedy
    match and the AP sweep are independent per category.  So a rank that holds
    the slice of its category block needs NO record exchange: it runs the
    ordinary single-GPU stages on the slice and sweeps its categories into the
    category-major tables.  The result then travels in ONE in-place
    ``all_gather_into_tensor`` of run-length packed chunks (csrc/exchange.hip:
    a row's 101 recall columns hold at most min(num_gt, 100) + 1 distinct
    values, and which columns coincide follows from num_gt alone), and every
    rank expands the chunks straight into the reference layout.

    Building the evaluator is collective: the chunk capacity is the largest
    block's level count, derived once from the all-gathered num_gt (the ground
    truth of a plan does not change between passes).
    """

    def __init__(self, dp, ws, rank, world, backend, group=None):
        self.dp, self.ws, self.rank, self.world = dp, ws, rank, world
        self.be, self.group = backend, group
        dev = dp.device
        K, R = dp.n_cat, dp.n_rng
        self.k0, self.k1, self.Kb = category_block(K, rank, world)
        Kb = self.Kb
        # own block of the category-major tables, addressed by global row
        self.val = torch.zeros((K, R, N_THR, N_REC), dtype=torch.float64, device=dev)
        self.rec = torch.zeros((K, R, N_THR), dtype=torch.float64, device=dev)
        self.num_gt = torch.zeros((K, R), dtype=torch.int32, device=dev)
        self.precision = torch.empty((N_THR, N_REC, K, R), dtype=torch.float64,
                                     device=dev)
        self.recall = torch.empty((N_THR, K, R), dtype=torch.float64, device=dev)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        self.xws = torch.empty(backend.exchange_workspace(Kb, R, world),
                               dtype=torch.uint8, device=dev)
        # ---- capacity: levels of the largest block (collective, once)
        backend.ranges(dp, ws)
        table = torch.zeros((Kb * world, R), dtype=torch.int32, device=dev)
        lo, hi = rank * Kb, min((rank + 1) * Kb, K)
        table[lo:hi].copy_(ws.num_gt[lo:hi])
        dist.all_gather_into_tensor(table, table[rank * Kb:(rank + 1) * Kb].clone(),
                                    group=group)
        totals = torch.zeros(world, dtype=torch.int64, device=dev)
        backend.exchange_sizes(Kb, R, world, table, totals, self.xws)
        self.capacity = int(totals.max().item())
        self.chunk_bytes = backend.exchange_chunk_bytes(Kb, R, self.capacity)
        self.chunks = torch.zeros(world * self.chunk_bytes, dtype=torch.uint8,
                                  device=dev)

    def compute(self, aux=None):
        """Local stages (no collective): ranges, sort, [3D IoU], match, sweep,
        pack of the own block into its chunk.  `aux`: a second stream for the
        stages that do not depend on each other."""
        dp, ws, be = self.dp, self.ws, self.be
        if aux is not None:
            be.local_head(dp, ws, aux)
        else:
            be.ranges(dp, ws)
            be.sort_local(dp, ws)
            be.track_iou(dp, ws)
            be.match_local(dp, ws)
        self.sweep()

    def sweep(self):
        """AP sweep of the own categories + pack into the own chunk."""
        dp, ws, be = self.dp, self.ws, self.be
        be.accumulate_compact(dp.n_dt, dp.n_cat, dp.n_rng, dp.t["cat_off"],
                              ws.matched, ws.ignored, ws.num_gt, self.k0,
                              self.k1, self.val, self.rec, ws.acc_ws,
                              ws.acc_bytes, dp.acc_hint)
        lo, hi = self.rank * self.chunk_bytes, (self.rank + 1) * self.chunk_bytes
        be.exchange_pack(dp.n_cat, dp.n_rng, self.Kb, self.world, self.rank,
                         ws.num_gt, self.val, self.rec, self.chunks[lo:hi],
                         self.capacity, self.overflow, self.xws)

    def gather(self):
        """The only collective of the pass."""
        lo, hi = self.rank * self.chunk_bytes, (self.rank + 1) * self.chunk_bytes
        dist.all_gather_into_tensor(self.chunks, self.chunks[lo:hi],
                                    group=self.group)

    def expand(self):
        """Chunks of all ranks -> reference layout."""
        dp = self.dp
        self.be.exchange_unpack(dp.n_cat, dp.n_rng, self.Kb, self.world,
                                self.chunks, self.capacity, self.num_gt,
                                self.precision, self.recall, self.overflow,
                                self.xws)

    def assemble(self):
        self.gather()
        self.expand()

    def step(self):
        self.compute()
        self.assemble()

    def check(self):
        """Host-side guard (synchronises): the capacity held."""
        if int(self.overflow.item()):
            raise _lib.TaoAmdError("exchange chunk overflow: the ground truth "
                                   "changed after the plan was built")


class CategoryPlan:
    """Both evaluators of one rank, category-partitioned.

    The image-level evaluator runs on the caller's stream, the track-level one
    on its own (plus one auxiliary stream each for the stages that are
    independent inside a pass), so the image-level all-gather travels while the
    track-level kernels still run; keeping the longer, image-level chain free
    of cross-stream waits took the one-rank step from 0.63 to 0.57 ms and a
    rank's share of an 8-rank job from 0.72 to 0.60 ms.  Five streams are busy
    at once (4 + RCCL's): run with GPU_MAX_HW_QUEUES >= 8, the
    default of 4 hardware queues per process makes them alias (bench.py sets
    it)."""

    def __init__(self, dpl, dpt, rank, world, device, backend=None, group=None):
        from . import engine
        self.device = torch.device(device)
        backend = backend or HipBackend()
        self.lvis = CategoryShardedEval(dpl, engine.Workspace(dpl), rank, world,
                                        backend, group)
        self.tao = CategoryShardedEval(dpt, engine.Workspace(dpt), rank, world,
                                       backend, group)
        self.streams = None
        if self.device.type == "cuda":
            self.streams = [torch.cuda.Stream(self.device) for _ in range(4)]

    def pair_frames(self):
        return int(self.tao.ws.pair_frames.item())

    def step(self):
        if self.streams is None:
            self.lvis.step()
            self.tao.step()
            return
        cur = torch.cuda.current_stream(self.device)
        # the image-level chain (the longer one) stays on the caller's stream,
        # only the track-level pass is forked and joined (engine.Overlap)
        st = self.streams[1]
        st.wait_stream(cur)
        self.lvis.compute(self.streams[2])
        self.lvis.assemble()
        with torch.cuda.stream(st):
            self.tao.compute(self.streams[3])
            self.tao.assemble()
        cur.wait_stream(st)


class ExchangePlan:
    """Both evaluators of one rank, unit-partitioned (`bench.py --shard unit`);
    the two passes run on their own HIP streams."""

    def __init__(self, dpl, dpt, rank, world, device, backend=None, group=None):
        from . import engine
        self.device = torch.device(device)
        backend = backend or HipBackend()
        self.lvis = ShardedEval(dpl, engine.Workspace(dpl), rank, world, backend,
                                group)
        self.tao = ShardedEval(dpt, engine.Workspace(dpt), rank, world, backend,
                               group)
        self.streams = None
        if self.device.type == "cuda":
            self.streams = [torch.cuda.Stream(self.device) for _ in range(2)]

    def pair_frames(self):
        return int(self.tao.ws.pair_frames.item())

    def step(self):
        if self.streams is None:
            self.lvis.step()
            self.tao.step()
            return
        cur = torch.cuda.current_stream(self.device)
        st = self.streams[1]            # image level on the caller's stream
        st.wait_stream(cur)
        self.lvis.step()
        with torch.cuda.stream(st):
            self.tao.step()
        cur.wait_stream(st)


def step(plan):
    plan.step()


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
    per_dt = ["dt_score", "dt_flags", "dt_id", "dt_cat", "dt_cell", "dt_row"]
    per_gt = ["gt_flags", "gt_id", "gt_cat", "gt_cell", "gt_row"]
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
        elif k == "masks":        # run-length masks (segm): CSR per row
            f[k] = None if v is None else {"dt": v["dt"].slice(d0, d1),
                                           "gt": v["gt"].slice(g0, g1)}
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
