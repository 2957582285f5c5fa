"""Multi-GPU evaluation: one process per GPU, RCCL (``torch.distributed``
backend "nccl") over xGMI.  Two partitions of the path are implemented:

* BY CATEGORY (``CategoryPlan``, ``bench.py --shard category``): the match and the AP
  sweep are independent per category and the cell tables are category-major,
  so a rank evaluates a contiguous category block with no record exchange at
  all; the only collectives are the in-place all-gathers that assemble the
  category-major result tables.  This is the natural mode when one
  prediction file is evaluated on several GPUs.
* BY UNIT (``ExchangePlan``, the default of bench.py and what the CLI runs
  under torchrun, evaluation/_dist.py): every rank holds the detections of its own
  images / videos (e.g. produced by data-parallel inference) and the records
  meet at the category owners -- described next.

Units (images / videos) sharded across ranks:

Why an exchange is needed at all: IoU + greedy match is
independent per cell, but AP needs every category's detections in ONE global
score order (reference lvis_amodal/eval.py:353-361).  A histogram all-reduce
is not enough for exact AP; the per-detection TP/ignore words have to meet.

Per evaluator pass and rank (round 5: the exchange in two messages, the
second in category blocks -- nothing of it waits behind the match that could
run beside it):

  1. range masks; ``all_reduce(sum)`` of num_gt[K, n_rng]        (tiny; side stream)
  2. local segment sort (the rank's tables are category-major): a detection's
     sorted place -- where its record lies in everything that follows, so what
     a rank sends to an owner is the runs of the owner's categories, each in
     (-score, concatenation) order.  Category k is owned by rank
     k // ceil(K / world): contiguous blocks.
  3. the SCORES at their sorted place -> ``all_to_all`` #1 (8 bytes a record),
     on the communication stream while the 3D IoU and the match run; from the
     scores alone the owner works out every record's row in the reference's
     order (taoamd_exchange_positions: a record's row = its place in its run +
     the records of the other sources' runs that precede it, one binary search
     per other source; ranks hold ascending, disjoint unit ranges and the
     sources are taken in rank order, so "lower rank first on ties" == the
     reference's concatenation order and its stable sort is reproduced
     exactly) -- on a side stream, beside the match.
  4. the match, in B phases: phase j matches the cells of the j-th sub-block
     of EVERY owner's category block and writes the (matched, ignored) pairs at
     their sorted place; ``all_to_all`` #2.j ships exactly those rows (16 bytes
     a record and combo word; every link busy in every phase) while phase
     j + 1 is matched.  A rank's own block never travels.
  5. owner: one scatter of the received pairs to their rows
     (taoamd_exchange_place) -- the paired table the sweep streams --, sweep
     of its categories (taoamd_accumulate_compact), result rows run-length
     packed (taoamd_exchange_pack)
  6. ONE in-place ``all_gather_into_tensor`` of the packed chunks, expanded
     straight into the reference layout (taoamd_exchange_unpack) -- the same
     tail as the category partition.

With one rank there is nothing to exchange or merge: the sweep reads the rows
where the match wrote them.

The collective plumbing is backend-agnostic: ``tests/test_dist_gloo.py`` runs
this very module with world_size 2 on CPU tensors over gloo, with the oracle
standing in for the kernels.
"""
import os

# (the by-video step keeps 4 compute streams + RCCL's busy: with the default of
# 4 hardware queues per process they alias and serialise -- 1.25 -> 0.94 ms/step
# with GPU_MAX_HW_QUEUES=8.  The HIP runtime reads it when it starts: whoever
# starts a multi-GPU process sets it before the first device call --
# evaluation/_dist.init_from_env, bench.py -- importing this module does not
# touch the environment.)

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
        # with the frame-order guard (engine.stage_iou_guard): a rank's tracks
        # are its own, so are the pairs to recompute -- on the device, no host
        # round trip (the host half only for ids Python hashes differently)
        self.engine.stage_track_iou_guarded(dp, ws)
        if dp.guard_flat is not None:
            ws.guarded_pairs = self.engine.apply_iou_guard(dp, ws)

    def scores_at_place(self, dp, ws, out):
        """out[sorted place of detection i] = bits of its score: the first
        message of the exchange (after sort_local)."""
        if dp.n_dt:
            _lib.check(self.lib.taoamd_exchange_scores(
                dp.n_dt, _ptr(ws.dst), _ptr(dp.t["dt_score"]), _ptr(out), self._s()),
                "taoamd_exchange_scores")

    def match_rows(self, dp, ws, phase=None):
        """The match of one phase's cells (None: all), rows at their sorted
        place in ws.rows.  `phase`: {"groups": (first, count) in
        phase["groups_dev"], "singles": ... in phase["singles_dev"]}."""
        if phase is None:
            return self.engine.stage_match(dp, ws)
        self.engine.stage_match(
            dp, ws, groups=(phase["groups_dev"], phase["groups"][0], phase["groups"][1]),
            singles=(phase["singles_dev"], phase["singles"][0], phase["singles"][1]))

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

    def positions(self, n_recv, world, block_cats, scores, own_scores, own, src_base,
                  run_off, cat_base, pos):
        _lib.check(self.lib.taoamd_exchange_positions(
            n_recv, world, block_cats, _ptr(scores), _ptr(own_scores), own,
            _ptr(src_base), _ptr(run_off), _ptr(cat_base), _ptr(pos), self._s()),
            "taoamd_exchange_positions")

    def place(self, n_recv, world, n_words, rows, own_rows, own, src_base, pos, out):
        _lib.check(self.lib.taoamd_exchange_place(
            n_recv, world, n_words, _ptr(rows), _ptr(own_rows), own, _ptr(src_base),
            _ptr(pos), _ptr(out), self._s()), "taoamd_exchange_place")

    def accumulate_compact(self, n, n_cat, n_rng, cat_off, matched, ignored,
                           num_gt, k0, k1, val, rec, ws_buf, ws_bytes,
                           max_segment=0, chunked=False):
        """`chunked`: the chunked kernels whatever the sweep mode (a pass whose
        look-back gave up is swept again, ShardedEval.check)."""
        fn = self.lib.taoamd_accumulate_compact_chunked if chunked \
            else self.lib.taoamd_accumulate_compact
        _lib.check(fn(
            n, n_cat, n_rng, _ptr(cat_off), _ptr(matched), _ptr(ignored),
            _ptr(num_gt), k0, k1, max_segment, _ptr(val), _ptr(rec),
            _ptr(ws_buf), ws_bytes, self._s()), "taoamd_accumulate_compact")

    def sweep_flag(self, ws_buf):
        """The one-pass sweep's error flag of the workspace a pass swept on
        (synchronises)."""
        return self.engine.sweep_flag(ws_buf, ws_buf.device)

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
                      rec, chunk, capacity, overflow, xws, maps_ready=False):
        _lib.check(self.lib.taoamd_exchange_pack(
            n_cat, n_rng, block_cats, world, rank, _ptr(num_gt), _ptr(val),
            _ptr(rec), _ptr(chunk), capacity, _ptr(overflow), _ptr(xws),
            xws.numel(), int(maps_ready), self._s()), "taoamd_exchange_pack")

    def exchange_unpack(self, n_cat, n_rng, block_cats, world, chunks, capacity,
                        num_gt, precision, recall, overflow, xws, maps_ready=False):
        _lib.check(self.lib.taoamd_exchange_unpack(
            n_cat, n_rng, block_cats, world, _ptr(chunks), capacity,
            _ptr(num_gt), _ptr(precision), _ptr(recall), _ptr(overflow),
            _ptr(xws), xws.numel(), int(maps_ready), self._s()),
            "taoamd_exchange_unpack")


PHASES = 4      # category sub-blocks per owner of the rows exchange (message #2)


class _Done:
    """A finished exchange (the synchronous, host-staged form)."""

    def wait(self):
        return True


def exchange_pieces(out_list, in_list, group=None):
    """``all_to_all`` of per-rank pieces (piece r of `in_list` goes to rank r,
    piece s of `out_list` comes from rank s; views into larger buffers, empty
    where nothing travels).  One GPU per rank: RCCL, asynchronous -- returns
    the work handle, ``wait()`` makes the CURRENT STREAM wait, the host never
    blocks.  gloo has no list form and moves no device tensors: the pieces are
    packed, staged on the host and exchanged with ``all_to_all_single``
    (ranks of a job sharing one GPU, the CPU tests)."""
    if dist.get_backend(group) != "gloo":
        return dist.all_to_all(out_list, in_list, group=group, async_op=True)
    inp = torch.cat([t.reshape(-1) for t in in_list]).cpu() if in_list else torch.empty(0)
    out = torch.empty(sum(t.numel() for t in out_list), dtype=inp.dtype)
    dist.all_to_all_single(out, inp, [t.numel() for t in out_list],
                           [t.numel() for t in in_list], group=group)
    at = 0
    for t in out_list:
        if t.numel():
            t.copy_(out[at:at + t.numel()].view(t.shape))
        at += t.numel()
    return _Done()


class ShardedEval:
    """One evaluator (LVIS or TAO side) of one rank when the problem is
    partitioned BY UNIT: the rank holds the cell tables of its own images /
    videos (category-major, flatten.py) and owns a contiguous category block
    of the result."""

    def __init__(self, dp, ws, rank, world, backend, group=None, phases=None):
        self.dp, self.ws, self.rank, self.world = dp, ws, rank, world
        self.be, self.group = backend, group
        dev = dp.device
        K, nw, R = dp.n_cat, dp.n_words, dp.n_rng
        self.k0, self.k1, self.Kb = category_block(K, rank, world)
        Kb = self.Kb
        # ---- who sends what: records per (owner, category of its block)
        cnt = np.zeros(Kb * world, dtype=np.int64)
        cnt[:K] = np.diff(dp.cat_off_host)
        send = torch.from_numpy(cnt).to(dev)                  # [world * Kb]
        recv = torch.empty_like(send)                         # [source][kb]
        all_to_all(recv, send, group=group)
        rc = recv.cpu().numpy().reshape(world, Kb)
        self.send_counts = cnt.reshape(world, Kb).sum(1).tolist()
        self.recv_counts = rc.sum(1).tolist()
        self.n_recv = int(rc.sum())
        # the rank's own block never travels: it is read where the sort / the
        # match wrote it
        self.own_at = sum(self.send_counts[:rank])
        self.n_wire = self.n_recv - self.recv_counts[rank]
        src_base = np.zeros(world + 1, dtype=np.int64)
        np.cumsum(rc.sum(1), out=src_base[1:])
        run_off = np.zeros((world, Kb + 1), dtype=np.int64)
        np.cumsum(rc, axis=1, out=run_off[:, 1:])
        cat_base = np.zeros(Kb + 1, dtype=np.int64)
        np.cumsum(rc.sum(0), out=cat_base[1:])
        cat_off = np.zeros(K + 1, dtype=np.int64)
        hi = min(self.k0 + Kb, K)
        cat_off[self.k0:hi + 1] = cat_base[:hi - self.k0 + 1]
        cat_off[hi + 1:] = self.n_recv
        self.max_segment = int(rc.sum(0).max()) if self.n_recv else 0
        self.src_base = torch.from_numpy(src_base).to(dev)
        self.run_off = torch.from_numpy(run_off).to(dev)
        self.cat_base = torch.from_numpy(cat_base).to(dev)
        self.cat_off = torch.from_numpy(cat_off.astype(np.int32)).to(dev)
        # ---- buffers.  Message #1: scores at their sorted place; message #2:
        # the rows of ws.rows (where the match writes them: the send buffer IS
        # the plain pass's row table).  `pos`: row of a received record,
        # `rows`: the paired table the sweep reads -- with one rank the match's.
        i64 = torch.int64
        self.send_score = torch.zeros(max(dp.n_dt, 1), dtype=i64, device=dev)
        self.recv_score = torch.zeros(max(self.n_wire, 1), dtype=i64, device=dev)
        self.recv_rows = torch.zeros((max(self.n_wire, 1), nw, 2), dtype=i64, device=dev)
        self.pos = torch.zeros(max(self.n_recv, 1), dtype=torch.int32, device=dev)
        self.rows = ws.rows if world == 1 else \
            torch.zeros((max(self.n_recv, 1), nw, 2), dtype=i64, device=dev)
        self.matched, self.ignored = self.rows[..., 0], self.rows[..., 1]
        # ---- the pieces of both messages (element ranges of the buffers)
        wire_base = src_base[:-1] - np.where(np.arange(world) > rank,
                                             self.recv_counts[rank], 0)
        cs = np.zeros(Kb * world + 1, dtype=np.int64)
        np.cumsum(cnt, out=cs[1:])
        self._score_pieces = (
            [(int(cs[r * Kb]), int(cs[(r + 1) * Kb])) if r != rank else (0, 0)
             for r in range(world)],
            [(int(wire_base[s_]), int(wire_base[s_] + rc[s_].sum())) if s_ != rank else (0, 0)
             for s_ in range(world)])
        B = phases if phases is not None else (1 if world == 1 else PHASES)
        B = max(1, min(B, Kb))
        edges = [Kb * j // B for j in range(B + 1)]
        self.phases = []
        for j in range(B):
            e0, e1 = edges[j], edges[j + 1]
            # (the phase's categories: sub-block j = [e0, e1) of every owner's block)
            self.phases.append({
                "send": [(int(cs[r * Kb + e0]), int(cs[r * Kb + e1])) if r != rank else (0, 0)
                         for r in range(world)],
                "recv": [(int(wire_base[s_] + run_off[s_, e0]), int(wire_base[s_] + run_off[s_, e1]))
                         if s_ != rank else (0, 0) for s_ in range(world)]})
        self._plan_match_phases(edges)
        self._side = torch.cuda.Stream(dev) if dev.type == "cuda" and world > 1 else None
        lib = _lib.load()
        self.acc_bytes = lib.taoamd_accumulate_workspace(self.n_recv, K, R)
        self.acc_ws = torch.empty(max(self.acc_bytes, 256), dtype=torch.uint8,
                                  device=dev)
        self.val = torch.zeros((K, R, N_THR, N_REC), dtype=torch.float64, device=dev)
        self.rec = torch.zeros((K, R, N_THR), dtype=torch.float64, device=dev)
        self.num_gt = torch.zeros((K, R), dtype=torch.int32, device=dev)
        self.num_gt_out = torch.zeros((K, R), dtype=torch.int32, device=dev)
        self.precision = torch.empty((N_THR, N_REC, K, R), dtype=torch.float64,
                                     device=dev)
        self.recall = torch.empty((N_THR, K, R), dtype=torch.float64, device=dev)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        self.xws = torch.empty(backend.exchange_workspace(Kb, R, world),
                               dtype=torch.uint8, device=dev)
        # ---- chunk capacity: levels of the largest block, from the global
        # num_gt (the ground truth of a plan does not change between passes)
        backend.ranges(dp, ws)
        self.num_gt.copy_(ws.num_gt)
        dist.all_reduce(self.num_gt, group=group)
        self.table = torch.zeros((Kb * world, R), dtype=torch.int32, device=dev)
        self.table[:K].copy_(self.num_gt)
        self.totals = torch.zeros(world, dtype=torch.int64, device=dev)
        backend.exchange_sizes(Kb, R, world, self.table, self.totals, self.xws)
        self.capacity = int(self.totals.max().item())
        self.chunk_bytes = backend.exchange_chunk_bytes(Kb, R, self.capacity)
        self.chunks = torch.zeros(world * self.chunk_bytes, dtype=torch.uint8,
                                  device=dev)

    def _plan_match_phases(self, edges):
        """The match's launch plan cut into the phases: the run descriptors
        (groups) and single cells permuted phase-major, a phase = one slice of
        each.  A run belongs to the EARLIEST phase among its cells' categories:
        its rows are then ready early, never late.  (The tables are
        category-major and a phase's categories are sub-block j of every
        owner's block, so inside one owner's block the phase follows from the
        run's first category; a run that straddles two owners' blocks holds
        categories of the next owner's FIRST sub-block: phase 0.)"""
        dp = self.dp
        if len(self.phases) == 1 or dp.n_dt == 0:
            for ph in self.phases:
                ph["groups"] = ph["singles"] = None
            return
        Kb = self.Kb
        edges = np.asarray(edges)

        def phase_of(first_dt, last_dt):
            c0 = np.searchsorted(dp.cat_off_host, first_dt, "right") - 1
            c1 = np.searchsorted(dp.cat_off_host, np.maximum(last_dt, first_dt), "right") - 1
            ph = np.searchsorted(edges, c0 % Kb, "right") - 1
            return np.where(c0 // Kb != c1 // Kb, 0, ph)
        dev = dp.device
        gh, sh = dp.groups_host, dp.singles_host
        pg = phase_of(gh[:, 0], gh[:, 0] + gh[:, 1] - 1) if len(gh) else np.zeros(0, np.int64)
        ps = phase_of(dp.singles_first_dt, dp.singles_first_dt) if len(sh) \
            else np.zeros(0, np.int64)
        og, os_ = np.argsort(pg, kind="stable"), np.argsort(ps, kind="stable")
        g_dev = torch.from_numpy(np.ascontiguousarray(gh[og]) if len(gh)
                                 else np.zeros((1, 4), np.int32)).to(dev)
        s_dev = torch.from_numpy(np.ascontiguousarray(sh[os_]) if len(sh)
                                 else np.zeros(1, np.int32)).to(dev)
        gb = np.searchsorted(pg[og], np.arange(len(self.phases) + 1))
        sb = np.searchsorted(ps[os_], np.arange(len(self.phases) + 1))
        for j, ph in enumerate(self.phases):
            ph["groups_dev"], ph["singles_dev"] = g_dev, s_dev
            ph["groups"] = (int(gb[j]), int(gb[j + 1] - gb[j]))
            ph["singles"] = (int(sb[j]), int(sb[j + 1] - sb[j]))

    def _global_num_gt(self):
        """num_gt of all rows (one all-reduce) and, from it, the run maps and
        level offsets the result exchange needs -- known at the head of the
        pass, beside the sort and the match, not after the sweep."""
        self.num_gt.copy_(self.ws.num_gt)
        dist.all_reduce(self.num_gt, group=self.group)
        self.table[:self.dp.n_cat].copy_(self.num_gt)
        self.be.exchange_sizes(self.Kb, self.dp.n_rng, self.world, self.table,
                               self.totals, self.xws)

    def _pieces(self, buf, ranges):
        return [buf[a:b] for a, b in ranges]

    def _own(self, buf):
        """The rank's own block inside a send-side buffer (it never travels).
        A rank without a detection in its own or any later owner's block has
        `own_at` == the buffer's length: the empty slice there has a NULL
        data pointer, which the C entries refuse whenever records arrive --
        the buffer's base stands in (no element of an empty block is read)."""
        return buf[self.own_at:] if self.own_at < buf.shape[0] else buf

    def _exchange_scores(self, send=None, recv=None):
        """Message #1 (also what carries the detections' ids for
        eval['dt_pointers']: owner_rows)."""
        send = self.send_score if send is None else send
        recv = self.recv_score if recv is None else recv
        return exchange_pieces(self._pieces(recv, self._score_pieces[1]),
                               self._pieces(send, self._score_pieces[0]), self.group)

    def step(self, aux=None):
        """`aux`: a second stream for the range masks and the num_gt
        all-reduce, which only the sweep needs (sort, 3D IoU and match run
        beside them)."""
        dp, ws, be = self.dp, self.ws, self.be
        cur = None
        if aux is not None:
            cur = torch.cuda.current_stream(dp.device)
            aux.wait_stream(cur)
            with torch.cuda.stream(aux):
                be.ranges(dp, ws)
                self._global_num_gt()
        else:
            be.ranges(dp, ws)
            self._global_num_gt()
        be.sort_local(dp, ws)
        lone = self.world == 1
        if not lone:
            # message #1 leaves as soon as the places are known; the rows of
            # the records are worked out from it beside the 3D IoU / the match
            be.scores_at_place(dp, ws, self.send_score)
            scores_in = self._exchange_scores()
            side = self._side
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(dp.device))
            with (torch.cuda.stream(side) if side is not None else _null()):
                scores_in.wait()
                be.positions(self.n_recv, self.world, self.Kb, self.recv_score,
                             self._own(self.send_score), self.rank, self.src_base,
                             self.run_off, self.cat_base, self.pos)
        be.track_iou(dp, ws)
        if cur is not None:
            cur.wait_stream(aux)         # the match reads the range masks
        arrivals = []
        for ph in self.phases:
            be.match_rows(dp, ws, ph if ph["groups"] is not None else None)
            if not lone:
                # message #2.j: this phase's rows, while the next phase is matched
                arrivals.append(exchange_pieces(self._pieces(self.recv_rows, ph["recv"]),
                                                self._pieces(ws.rows, ph["send"]),
                                                self.group))
        if not lone:
            for w in arrivals:
                w.wait()
            if self._side is not None:
                torch.cuda.current_stream(dp.device).wait_stream(self._side)
            be.place(self.n_recv, self.world, dp.n_words, self.recv_rows,
                     self._own(ws.rows), self.rank, self.src_base, self.pos, self.rows)
        self._sweep()
        self._publish()

    def _sweep(self, chunked=False):
        """The owner's sweep of the merged rows of its category block."""
        dp = self.dp
        self.be.accumulate_compact(self.n_recv, dp.n_cat, dp.n_rng, self.cat_off,
                                   self.matched, self.ignored, self.num_gt, self.k0,
                                   self.k1, self.val, self.rec, self.acc_ws,
                                   self.acc_bytes, self.max_segment, chunked=chunked)

    def _publish(self):
        """Result exchange: pack the own block, all_gather, expand."""
        dp, be = self.dp, self.be
        lo, hi = self.rank * self.chunk_bytes, (self.rank + 1) * self.chunk_bytes
        be.exchange_pack(dp.n_cat, dp.n_rng, self.Kb, self.world, self.rank,
                         self.num_gt, self.val, self.rec, self.chunks[lo:hi],
                         self.capacity, self.overflow, self.xws, maps_ready=True)
        dist.all_gather_into_tensor(self.chunks, self.chunks[lo:hi],
                                    group=self.group)
        be.exchange_unpack(dp.n_cat, dp.n_rng, self.Kb, self.world, self.chunks,
                           self.capacity, self.num_gt_out, self.precision,
                           self.recall, self.overflow, self.xws, maps_ready=True)

    def check(self):
        """Host-side guard of the last step (synchronises; COLLECTIVE): the
        exchange capacity held, and no rank's sweep gave up a look-back -- if
        one did, every rank sweeps its block again with the chunked kernels
        (the merged rows are still in place) and the result exchange is
        repeated, so the tables are right when this returns.  Under the
        evaluation constants of the step."""
        if int(self.overflow.item()):
            raise _lib.TaoAmdError("exchange chunk overflow: the ground truth "
                                   "changed after the plan was built")
        if any_rank(self.be.sweep_flag(self.acc_ws), self.dp.device, self.group):
            _warn_sweep()
            self._sweep(chunked=True)
            self._publish()
            if self.dp.device.type == "cuda":
                torch.cuda.synchronize(self.dp.device)
            self.sweep_recovered = getattr(self, "sweep_recovered", 0) + 1
            if int(self.overflow.item()):
                raise _lib.TaoAmdError("exchange chunk overflow")

    def owner_rows(self, ids):
        """After step(): the rows of this rank's category block as the sweep
        saw them -- (ids, matched, ignored, cat_base) with `ids` (int64 device
        tensor, one per detection of this rank) taken through the very exchange
        the scores went through (message #1's pieces, the ids at the sorted
        places) and put at the rows the pass worked out for the records
        (`pos`).  Collective.  The source of eval['dt_pointers'] in a multi-GPU
        run (T/eval.py:575-584)."""
        dp, ws = self.dp, self.ws
        n = self.n_recv
        send = torch.zeros_like(self.send_score)
        if dp.n_dt:
            send[ws.dst[:dp.n_dt].long()] = ids
        if self.world == 1:
            out = send[:n]
        else:
            recv = torch.zeros_like(self.recv_score)
            self._exchange_scores(send, recv).wait()
            below = int(sum(self.recv_counts[:self.rank]))
            own = send[self.own_at:self.own_at + self.recv_counts[self.rank]]
            rec = torch.cat([recv[:below], own, recv[below:self.n_wire]])
            out = torch.zeros(max(n, 1), dtype=torch.int64, device=send.device)
            out[self.pos[:n].long()] = rec[:n]
        return (out[:n].cpu().numpy(),
                self.matched[:n].contiguous().cpu().numpy().view(np.uint64),
                self.ignored[:n].contiguous().cpu().numpy().view(np.uint64),
                self.cat_base.cpu().numpy())


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def any_rank(flag, device, group=None):
    """True on every rank if `flag` is set on any (one all-reduce)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(flag)
    t = torch.tensor([int(bool(flag))], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(t.item())


def _warn_sweep():
    import logging
    logging.getLogger("tao_amodal_amd").warning(
        "the one-pass sweep's look-back between workgroups timed out on a rank: "
        "the pass is swept again with the chunked kernels")


def all_to_all(output, input, output_split_sizes=None, input_split_sizes=None, group=None):
    """``dist.all_to_all_single`` of the record exchange.  One GPU per rank:
    RCCL, device buffers over xGMI.  When the ranks of a job share one GPU (a
    development box, the multi-rank tests on a one-GPU box) they talk over
    gloo, which moves no device tensors through all_to_all: staged on the host."""
    if output.is_cuda and dist.get_backend(group) == "gloo":
        o = torch.empty(output.shape, dtype=output.dtype)
        dist.all_to_all_single(o, input.cpu(), output_split_sizes, input_split_sizes,
                               group=group)
        output.copy_(o)
        return
    dist.all_to_all_single(output, input, output_split_sizes, input_split_sizes,
                           group=group)


def gather_visit_universe(gt, device, group=None):
    """Image ids of the WHOLE ground truth when every rank holds the share of
    its own videos (ranks hold ascending video ranges): the CPython set whose
    iteration order fixes the visiting order of the track level
    (flatten.tao_gt_side) is built from all of them, not from a rank's share."""
    from .flatten import video_images
    own = torch.from_numpy(np.ascontiguousarray(video_images(gt))).to(device)
    world = dist.get_world_size(group)
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    sizes[dist.get_rank(group)] = own.numel()
    dist.all_reduce(sizes, group=group)
    m = int(sizes.max().item())
    pad = torch.zeros(m, dtype=torch.int64, device=device)
    pad[:own.numel()] = own
    out = torch.empty(world * m, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, pad, group=group)
    out, sizes = out.cpu().numpy().reshape(world, m), sizes.cpu().numpy()
    return np.concatenate([out[r, :sizes[r]] for r in range(world)])


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

    def sweep(self, chunked=False):
        """AP sweep of the own categories + pack into the own chunk."""
        dp, ws, be = self.dp, self.ws, self.be
        be.accumulate_compact(dp.n_dt, dp.n_cat, dp.n_rng, dp.t["cat_off"],
                              ws.matched, ws.ignored, ws.num_gt, self.k0,
                              self.k1, self.val, self.rec, ws.acc_ws,
                              ws.acc_bytes, dp.acc_hint, chunked=chunked)
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

    def owner_rows(self, ids):
        """ShardedEval.owner_rows for the category partition: the rows of this
        rank's block are its own pass's (no exchange)."""
        dp, ws = self.dp, self.ws
        n = dp.n_dt
        order = ws.order[:n].long()
        hi = min(self.k0 + self.Kb, dp.n_cat)
        base = np.asarray(dp.cat_off_host[self.k0:hi + 1], dtype=np.int64)
        base = np.concatenate([base, np.full(self.Kb + 1 - len(base), base[-1] if len(base) else 0)])
        return (ids[order].cpu().numpy(), ws.matched[:n].cpu().numpy().view(np.uint64),
                ws.ignored[:n].cpu().numpy().view(np.uint64), base - (base[0] if len(base) else 0))

    def check(self):
        """Host-side guard of the last step (synchronises; COLLECTIVE): the
        capacity held and no rank's look-back gave up (ShardedEval.check: the
        block is swept again with the chunked kernels and exchanged again)."""
        if int(self.overflow.item()):
            raise _lib.TaoAmdError("exchange chunk overflow: the ground truth "
                                   "changed after the plan was built")
        if any_rank(self.be.sweep_flag(self.ws.acc_ws), self.dp.device, self.group):
            _warn_sweep()
            self.sweep(chunked=True)
            self.assemble()
            if self.dp.device.type == "cuda":
                torch.cuda.synchronize(self.dp.device)
                self.engine_prepare()
            self.sweep_recovered = getattr(self, "sweep_recovered", 0) + 1

    def engine_prepare(self):
        # (the unprepared chunked pass overwrote the workspace's plan)
        from . import engine
        engine.prepare_sweep(self.dp, self.ws)


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
    """Both evaluators of one rank, unit-partitioned (`bench.py --shard unit`):
    the image-level pass on the caller's stream, the track-level one on its
    own, one auxiliary stream each for the range masks + num_gt all-reduce."""

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
            self.streams = [torch.cuda.Stream(self.device) for _ in range(4)]

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
        self.lvis.step(self.streams[2])
        with torch.cuda.stream(st):
            self.tao.step(self.streams[3])
        cur.wait_stream(st)


def step(plan):
    plan.step()


# --------------------------------------------------------------------------
# cutting a cell range out of one flattened problem (category shards)
# --------------------------------------------------------------------------
def shard_flat(flat, c0, c1):
    """The sub-problem made of cells [c0, c1)."""
    from .flatten import Flat
    if hasattr(flat, "to_host"):        # a device-built flat: its tables downloaded
        flat = flat.to_host()
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
