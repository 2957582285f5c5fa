"""One evaluation on several GPUs behind the plugin surface:

    torchrun --nproc_per_node=N tools/eval_on_tao_amodal.py --track_result ...

(and ``LVISEval(..., dist=ctx)`` / ``TaoEval(..., dist=ctx)`` in the class
API).  One process per GPU, RCCL over xGMI (``torch.distributed`` backend
"nccl"); the partition is the one BASELINE.json names -- by video -- with the
per-detection records meeting at the category owners (``dist.ShardedEval``).

What a rank holds:

  image level   the images of one contiguous block of the SORTED IMAGE IDS, the
                annotations on them and the predictions for them;
  track level   the videos of one contiguous block of the sorted video ids,
                their images / tracks / annotations and predictions.

Blocks of sorted ids, lower ids on lower ranks: the reference concatenates a
category's cells in sorted image / video id order before its stable score
sort (lvis_amodal/eval.py:343-361, tao_amodal/eval.py:498-518), and the owner
of a category merges the ranks' runs "lower rank first on ties".

Every rank reads the whole annotation file (it needs the id universe and the
category table) and converts only ITS share of the prediction list
(``DTColumns.from_file_native(part=rank)``: the structural scan of the file is
shared work, the number conversion is split); the records then travel to the
rank that owns their image / video in one all_to_all per level, arriving in
file order.  ``make_track_ids_unique`` (tools/eval_on_tao_amodal.py:44-66) is a
statement about the WHOLE list -- ids shared by two videos are renumbered in
order of first appearance -- so the ranks pool their (track, video) pairs with
the position of their first record and each applies the same renumbering.
"""
import os

import numpy as np


class Ctx:
    """Process group of one evaluation: the ranks, the device of this rank, a
    host-side (gloo) group for small numpy exchanges."""

    def __init__(self, rank, world, device, group=None, host_group=None, backend="nccl"):
        self.rank, self.world, self.device = rank, world, device
        self.group, self.host_group, self.backend = group, host_group, backend
        # set by shard_inputs: every rank holds the WHOLE inputs and the ranks
        # split the categories instead of the videos (annotation files in which
        # two records share an id)
        self.whole = False
        # True: accumulate() also assembles eval['dt_pointers'] (collective: the
        # rows of a category live on its owner rank and are gathered to all)
        self.pointers = False


def init_from_env():
    """Process group from the launcher's environment (torchrun).  One GPU per
    rank -> RCCL; fewer GPUs than ranks (a development box) -> the ranks share
    GPU 0 and talk over gloo, device tensors staged through the host."""
    # (read by the HIP runtime when it starts: before the first device call)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    n_gpu = torch.cuda.device_count()
    if n_gpu == 0:
        raise RuntimeError("tao_amodal_amd evaluates on AMD GPUs; none is visible")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    own_gpu = n_gpu >= local_world
    dev = torch.device("cuda", local if own_gpu else 0)
    torch.cuda.set_device(dev)
    if not dist.is_initialized():
        if own_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    backend = dist.get_backend()
    host = dist.new_group(backend="gloo") if backend != "gloo" else None
    return Ctx(rank, world, dev, None, host, backend)


# --------------------------------------------------------------------------
# host-side exchanges
# --------------------------------------------------------------------------
def _gather_arrays(arr, ctx):
    """Every rank's array (any length), in rank order."""
    import torch.distributed as dist
    out = [None] * ctx.world
    dist.all_gather_object(out, np.ascontiguousarray(arr), group=ctx.host_group)
    return out


def _route_rows(mat, owner, ctx):
    """Rows of `mat` (int64 [n, W]) sent to their owner rank; what arrives is
    ordered by source rank, each source's rows in its own order -- shares of a
    list cut in order arrive in list order."""
    import torch
    import torch.distributed as dist
    order = np.argsort(owner, kind="stable")
    send = np.ascontiguousarray(mat[order])
    counts = np.bincount(owner, minlength=ctx.world).astype(np.int64)
    got = torch.zeros(ctx.world, dtype=torch.int64)
    dist.all_to_all_single(got, torch.from_numpy(counts), group=ctx.host_group)
    recv_counts = got.tolist()
    n_in, W = int(sum(recv_counts)), mat.shape[1]
    on_dev = ctx.backend == "nccl"
    src = torch.from_numpy(send)
    dst = torch.empty((n_in, W), dtype=torch.int64)
    if on_dev:          # over xGMI: device buffers
        src = src.to(ctx.device)
        dst = dst.to(ctx.device)
    dist.all_to_all_single(dst, src, output_split_sizes=recv_counts,
                           input_split_sizes=counts.tolist(),
                           group=ctx.group if on_dev else ctx.host_group)
    return dst.cpu().numpy()


def block_owner(sorted_ids, ids, world):
    """Rank owning each id: the sorted universe cut into `world` contiguous
    blocks of equal count; -1 for an id outside the universe."""
    n = len(sorted_ids)
    pos = np.searchsorted(sorted_ids, ids)
    pos_c = np.minimum(pos, max(n - 1, 0))
    known = (n > 0) & (sorted_ids[pos_c] == ids) if n else np.zeros(len(ids), bool)
    # block r = positions [n * r // world, n * (r + 1) // world)
    bounds = np.array([n * r // world for r in range(1, world)], dtype=np.int64)
    owner = np.searchsorted(bounds, pos_c, side="right")
    return np.where(known, owner, -1)


def block_mask(sorted_ids, ids, rank, world):
    return block_owner(sorted_ids, ids, world) == rank


def unique_track_ids(dt, first, ctx):
    """make_track_ids_unique over the whole list from the ranks' shares.
    Returns (new track ids of this share, number of ids shared by videos)."""
    import torch
    import torch.distributed as dist
    tid, vid = np.asarray(dt.track_id), np.asarray(dt.video_id)
    if len(tid):
        pairs, pfirst = np.unique(np.stack([tid, vid], 1), axis=0, return_index=True)
        mine = np.concatenate([pairs, (first + pfirst)[:, None]], 1)
    else:
        mine = np.zeros((0, 3), dtype=np.int64)
    allp = np.concatenate(_gather_arrays(mine, ctx))
    top = torch.tensor([int(tid.max()) if len(tid) else 0], dtype=torch.int64)
    dist.all_reduce(top, op=dist.ReduceOp.MAX, group=ctx.host_group)
    top = max(int(top.item()), 0)
    if len(allp) == 0:
        return tid.copy(), 0
    # first position of every (track, video) pair in the whole list
    o = np.lexsort((allp[:, 2], allp[:, 1], allp[:, 0]))
    allp = allp[o]
    head = np.ones(len(allp), bool)
    head[1:] = (allp[1:, 0] != allp[:-1, 0]) | (allp[1:, 1] != allp[:-1, 1])
    allp = allp[head]
    t_first = np.ones(len(allp), bool)
    t_first[1:] = allp[1:, 0] != allp[:-1, 0]
    per_track = np.add.reduceat(np.ones(len(allp), np.int64), np.flatnonzero(t_first))
    clash = np.repeat(per_track > 1, per_track)
    if not clash.any():
        return tid.copy(), 0
    cp = allp[clash]
    cp = cp[np.argsort(cp[:, 2], kind="stable")]          # order of first appearance
    new_id = top + 1 + np.arange(len(cp))
    # look the share's records up among the renumbered pairs
    out = tid.copy()
    if len(tid):
        comp = cp[:, :2]
        q = np.stack([tid, vid], 1)
        _, inv = np.unique(np.concatenate([comp, q]), axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        new_of = np.full(int(inv.max()) + 1, -1, dtype=np.int64)
        new_of[inv[:len(comp)]] = new_id
        hit = new_of[inv[len(comp):]]
        out = np.where(hit >= 0, hit, tid)
    return out, int((per_track > 1).sum())


class Shares:
    """This rank's inputs of both levels."""

    def __init__(self, gt_lvis, dt_lvis, gt_tao, dt_tao, n_changed, total, whole=False):
        self.gt_lvis, self.dt_lvis = gt_lvis, dt_lvis
        self.gt_tao, self.dt_tao = gt_tao, dt_tao
        self.n_changed, self.total = n_changed, total
        self.whole = whole


def shard_inputs(gt, dt, first, ctx):
    """`gt`: the whole annotation file (GTColumns), `dt`: this rank's share of
    the prediction list starting at list position `first`.  Returns Shares."""
    import torch
    import torch.distributed as dist
    from ..columns import DTColumns
    # The reference keeps images, videos, tracks and annotations in dicts keyed
    # by id: of two entries with one id the last replaces the first -- also
    # when they sit in different videos (T/tao.py:131-160, L/lvis.py:34-61).
    # A split by video cannot reproduce that replacement across ranks.  Every
    # rank sees the annotation file whole and decides alike: such a file is
    # evaluated with every rank holding the WHOLE inputs (the table build of
    # flatten.py knows the dicts' semantics) and the ranks splitting the
    # CATEGORIES (dist.CategoryShardedEval: no record exchange; the match and
    # the sweep are independent per category).
    whole = any(len(np.unique(ids)) != len(ids)
                for ids in (gt.img_id, gt.vid_id, gt.trk_id, gt.ann_id))
    img_sorted = np.unique(gt.img_id)
    vid_sorted = np.unique(gt.vid_id)
    own_img = block_owner(img_sorted, np.asarray(dt.image_id), ctx.world)
    own_vid = block_owner(vid_sorted, np.asarray(dt.video_id), ctx.world)
    bad = torch.tensor([int((own_img < 0).sum()), int((own_vid < 0).sum()), len(dt)],
                       dtype=torch.int64)
    dist.all_reduce(bad, group=ctx.host_group)
    if int(bad[2]) == 0:
        raise IndexError("list index out of range")      # results.py:61 of both
    if int(bad[0]):
        raise AssertionError("Results do not correspond to current LVIS set.")
    if int(bad[1]):
        raise AssertionError("Results do not correspond to current Tao set.")
    new_tid, n_changed = unique_track_ids(dt, first, ctx)

    def pack(track_id):
        m = np.empty((len(dt), 10), dtype=np.int64)
        m[:, 9] = first + np.arange(len(dt))       # place in the file's list
        m[:, 0] = dt.image_id
        m[:, 1] = dt.category_id
        m[:, 2:6] = np.ascontiguousarray(dt.bbox, dtype=np.float64).view(np.int64)
        m[:, 6] = np.ascontiguousarray(dt.score, dtype=np.float64).view(np.int64)
        m[:, 7] = track_id
        m[:, 8] = dt.video_id
        return m

    def unpack(m):
        d = DTColumns(image_id=m[:, 0].copy(), category_id=m[:, 1].copy(),
                      bbox=np.ascontiguousarray(m[:, 2:6]).view(np.float64),
                      score=np.ascontiguousarray(m[:, 6]).view(np.float64),
                      track_id=m[:, 7].copy(), video_id=m[:, 8].copy())
        d.file_pos = m[:, 9].copy()
        return d
    # (the image level never looks at track ids: it gets the renumbered ones too)
    mat = pack(new_tid)
    if whole:
        # the shares are consecutive pieces of the list in rank order
        every = unpack(np.concatenate(_gather_arrays(mat, ctx)))
        ctx.whole = True
        return Shares(gt, every, gt, every, n_changed, int(bad[2]), whole=True)
    dt_l = unpack(_route_rows(mat, own_img, ctx))
    dt_t = unpack(_route_rows(mat, own_vid, ctx))
    gt_l = gt.select_images(block_mask(img_sorted, gt.img_id, ctx.rank, ctx.world))
    gt_t = gt.select_videos(block_mask(vid_sorted, gt.vid_id, ctx.rank, ctx.world))
    return Shares(gt_l, dt_l, gt_t, dt_t, n_changed, int(bad[2]))


# --------------------------------------------------------------------------
# the evaluator pass of one rank
# --------------------------------------------------------------------------
class DistRun:
    """What ``_core.GpuRun`` is to one GPU: evaluate() runs this rank's share
    through ranges / sort / [3D IoU] / match, the record exchange, the owner's
    sweep and the result exchange (dist.ShardedEval.step); accumulate() waits
    and downloads the assembled tables, identical on every rank.  The lazy
    per-cell views show the rank's own cells."""

    def __init__(self, flat, ctx, iou_3d_type="3d_iou", constants=None, dt=None,
                 max_dets=300, subset=False):
        """``dt`` / ``max_dets`` (image level): the share's prediction columns
        and the cut of LVISResults -- what the global ``id`` of a detection
        (its place in the WHOLE post-truncation list, L/results.py:73-84) is
        worked out from when ctx.pointers asks for eval['dt_pointers']."""
        import torch
        self.dt, self.max_dets = dt, max_dets
        self._pointers = None
        if subset and ctx.pointers and flat.kind == "lvis" and not ctx.whole:
            # the reference numbers the detections over the WHOLE list before
            # any subset is taken (L/results.py:75-84, eval.py:59-105); a rank
            # only sees the subset's part of its share (alike on every rank)
            raise NotImplementedError(
                "eval['dt_pointers'] of a multi-GPU run with params.img_ids "
                "restricted to a subset")
        from .. import dist as tdist, engine
        from ._core import applied, timed
        self.engine, self.torch = engine, torch
        self.flat, self.iou_3d_type, self.ctx = flat, iou_3d_type, ctx
        self.device = ctx.device
        # edited params (EvalConstants): one block of thresholds per pass
        self.constants = constants if constants is not None and not constants.default \
            else None
        if self.constants is not None and not self.constants.single:
            raise NotImplementedError(
                "a multi-GPU run takes up to %d IoU and %d recall thresholds"
                % (engine.N_THR, engine.N_REC))
        with timed("upload+plan"), applied(self.constants):
            if ctx.whole:
                # `flat` is the whole problem on every rank: this rank's
                # category block of its cells
                k0, k1, _ = tdist.category_block(len(flat.cat_ids), ctx.rank, ctx.world)
                self._shard = tdist.shard_by_category(flat, k0, k1)
                self.dp = engine.DeviceProblem(self._shard, self.device, iou_3d_type)
                self.ws = engine.Workspace(self.dp)
                self.sharded = tdist.CategoryShardedEval(
                    self.dp, self.ws, ctx.rank, ctx.world, tdist.HipBackend(), ctx.group)
            else:
                self.dp = engine.DeviceProblem(flat, self.device, iou_3d_type)
                self.ws = engine.Workspace(self.dp)
                self.sharded = tdist.ShardedEval(self.dp, self.ws, ctx.rank, ctx.world,
                                                 tdist.HipBackend(), ctx.group)
            torch.cuda.synchronize(self.device)
        self._detail = None
        self.near_threshold_pairs = 0
        self.precision = self.recall = None

    def evaluate(self):
        from ._core import applied, timed
        with timed("kernels"), applied(self.constants):
            self.sharded.step()

    def accumulate(self):
        from ._core import applied, timed
        with timed("kernels"):
            self.torch.cuda.synchronize(self.device)
            # (collective; the flag of the workspace that was actually swept:
            # a rank whose look-back gave up makes every rank sweep again)
            with applied(self.constants):
                self.sharded.check()
            self.near_threshold_pairs = self.engine.guarded_pairs(
                self.dp, self.ws, check_sweep=False)
        with timed("download"):
            self.precision = self.sharded.precision.cpu().numpy()
            self.recall = self.sharded.recall.cpu().numpy()
        if self.ctx.pointers:
            self._pointers = self._gather_pointers()
        c = self.constants
        if c is not None:
            # the caller's thresholds, in the caller's order (GpuRun._accumulate_blocks)
            t_idx, r_idx = c.thr_blocks[0][0], c.rec_blocks[0][0]
            p = np.empty((c.T, c.R) + self.precision.shape[2:])
            p[np.ix_(t_idx, r_idx)] = self.precision[:c.T, :c.R]
            r = np.empty((c.T,) + self.recall.shape[1:])
            r[t_idx] = self.recall[:c.T]
            if not c.rec_sorted:
                p[np.maximum.accumulate(p == 0, axis=1)] = 0
            self.precision, self.recall = p, r

    def thr_slots(self):
        c = self.constants
        if c is None:
            return list(range(self.engine.N_THR))
        slot = np.empty(c.T, dtype=np.int64)
        slot[c.thr_blocks[0][0]] = np.arange(c.T)
        return slot.tolist()

    def detail(self):
        if self._detail is None:
            from ._core import applied
            with applied(self.constants):
                self._detail = self.engine.evaluate_flat(
                    self.flat, self.device, detail=True, iou_3d_type=self.iou_3d_type)
        return self._detail

    # ------------------------------------------------ eval['dt_pointers']
    def _global_ids(self):
        """`id` of this rank's detections as the reference numbers them."""
        from .. import flatten
        flat = self._shard if self.ctx.whole else self.flat
        local = np.asarray(flat.dt_id, dtype=np.int64)
        if flat.kind == "tao" or self.ctx.whole or self.dt is None:
            return local            # track ids; ids of the whole list
        # image level: id = 1 + place in the post-truncation list, which groups
        # the kept boxes by image in the order the images FIRST APPEAR in the
        # file (L/results.py:75-84).  A share holds whole images in file order:
        # its local list has the same groups in the same relative order, so an
        # id moves by (global - local) first row of its image's group.
        dt = self.dt
        keep = flatten.limit_dets_per_image(dt, self.max_dets)
        img = np.asarray(dt.image_id)[keep]
        head = np.ones(len(keep), bool)
        head[1:] = img[1:] != img[:-1]
        starts = np.flatnonzero(head)
        counts = np.diff(np.r_[starts, len(keep)])
        first_row = np.asarray(dt.file_pos)[
            np.unique(np.asarray(dt.image_id), return_index=True)[1]]
        # (np.unique sorts by image id: bring the first rows into group order)
        by_id = np.argsort(img[starts], kind="stable")
        first_pos = np.empty(len(starts), dtype=np.int64)
        first_pos[by_id] = first_row
        mine = np.stack([first_pos, counts], 1) if len(starts) else np.zeros((0, 2), np.int64)
        every = _gather_arrays(mine, self.ctx)
        table = np.concatenate(every)
        o = np.argsort(table[:, 0], kind="stable")
        base = np.empty(len(table), dtype=np.int64)
        base[o] = np.cumsum(table[o, 1]) - table[o, 1]
        at = sum(len(e) for e in every[:self.ctx.rank])
        gbase = base[at:at + len(starts)]
        pos = local - 1
        run = np.searchsorted(starts, pos, "right") - 1
        return gbase[run] + (pos - starts[run]) + 1

    def _gather_pointers(self):
        """Collective: every rank ends with the whole sorted tables."""
        ids = self.torch.from_numpy(np.ascontiguousarray(self._global_ids())).to(self.device)
        rows = self.sharded.owner_rows(ids)
        every = [None] * self.ctx.world
        import torch.distributed as dist
        dist.all_gather_object(every, rows, group=self.ctx.host_group)
        K = self.dp.n_cat
        cat_off = np.zeros(K + 1, dtype=np.int64)
        at, k = 0, 0
        for ids_r, _m, _i, base in every:
            kb = min(len(base) - 1, K - k)
            cat_off[k:k + kb + 1] = at + base[:kb + 1]
            at += len(ids_r)
            k += kb
        cat_off[k:] = at
        return {"ids": np.concatenate([e[0] for e in every]),
                "matched": np.concatenate([e[1] for e in every]),
                "ignored": np.concatenate([e[2] for e in every]),
                "num_gt": self.sharded.num_gt.cpu().numpy(), "cat_off": cat_off}

    def pointer_tables(self):
        if self._pointers is None:
            raise NotImplementedError(
                "eval['dt_pointers'] in a multi-GPU run: a category's rows live "
                "on its owner rank; set ctx.pointers = True before run() and "
                "accumulate() gathers them on every rank (a collective)")
        return self._pointers
