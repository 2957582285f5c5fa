#!/usr/bin/env python
"""Drop-in for the reference's ``tools/eval_on_tao_amodal.py``: same three
flags, same stdout / log-file text, evaluation on an MI355X.

    python tools/eval_on_tao_amodal.py --track_result prediction.json \
        --output_log out/eval.log --annotation validation_lvis_v1.json

Flow (reference tools/eval_on_tao_amodal.py:155-165): image-level LVISEval on
the pair, its 25 lines on stdout and the 21-metric table + two ``copypaste:``
lines in the log; then ``make_track_ids_unique`` and the track-level TaoEval,
its 19 lines and the four ``TAO 3DmAP...`` lines in the log.  Both JSON files
are parsed once and shared by the two evaluators (the reference parses each of
them twice and deep-copies the ground truth twice).

Several GPUs: the same command under ``torchrun --nproc_per_node=N`` (one
process per GPU, RCCL).  The evaluation is split by video over the ranks
(tao_amodal_amd/evaluation/_dist.py); rank 0 prints the same text and writes
the same log file, the other ranks are silent.
"""
import argparse
import logging
import os
import sys
from pathlib import Path

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
# (idle OpenMP threads sleep instead of spinning: tao_amodal_amd/__init__.py)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import json  # noqa: E402

from tao_amodal_amd import flatten  # noqa: E402
from tao_amodal_amd.columns import DTColumns  # noqa: E402
from tao_amodal_amd.evaluation.lvis_amodal import LVIS, LVISEval, LVISResults  # noqa: E402
from tao_amodal_amd.evaluation.tao_amodal import Tao, TaoEval, TaoResults  # noqa: E402

DEFAULT_ANNOTATION = (
    "/compute/trinity-1-38/chengyeh/TAO/amodal_annotations/"
    "validation_with_freeform_amodal_boxes_Aug10_2022_oof_visibility_GTR_"
    "lvis_v1.json")

LVIS_METRICS = ["AP", "AP50", "AP75",
                "AP-HO", "AP50-HO", "AP75-HO",
                "AP-PO", "AP50-PO", "AP75-PO",
                "AP-HV", "AP50-HV", "AP75-HV",
                "AP-OOF", "AP50-OOF", "AP75-OOF",
                "AP-HP", "AP50-HP", "AP75-HP", "APr", "APc", "APf"]


def default_arg_parser(argv=None):
    parser = argparse.ArgumentParser(
        description=__doc__.split("\n")[0],
        formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--track_result", type=str, required=True)
    parser.add_argument("--output_log", type=str, required=True)
    parser.add_argument("--annotation", type=str, default=None)
    return parser.parse_args(argv)


def create_small_table(small_dict):
    """The one-row pipe table the reference gets from
    ``detectron2.utils.logger.create_small_table`` (tabulate, tablefmt="pipe",
    floatfmt=".3f", centred): header padded by two, ``str.center`` cells."""
    keys = list(small_dict.keys())
    vals = ["{:.3f}".format(v) for v in small_dict.values()]
    widths = [max(len(k) + 2, len(v)) for k, v in zip(keys, vals)]
    row = lambda cells: "|" + "|".join(  # noqa: E731
        " {:^{w}s} ".format(c, w=w) for c, w in zip(cells, widths)) + "|"
    sep = "|" + "|".join(":" + "-" * w + ":" for w in widths) + "|"
    return "\n".join([row(keys), sep, row(vals)])


def make_track_ids_unique(dt):
    """reference tools/eval_on_tao_amodal.py:44-66, on columns (in place)."""
    ask = getattr(dt, "track_clash_free", None)
    if ask is not None and ask():
        # columns made on the device (columns.DeviceDTColumns): the usual answer
        # -- no id is shared between videos, nothing to renumber -- comes from
        # there, without waiting for the host arrays
        return 0
    dt.track_id, n = flatten.make_track_ids_unique(dt)
    return n


# What a call's levels built -- tables, columns, evaluators: gigabytes of host
# arrays -- is let go OFF the caller's path: the levels park their objects here
# and main() hands the lot to a helper thread once the tables are printed
# (unmapping them where they fall out of scope was 0.1-0.2 s of the call).
_PARKED = []


def _let_go(parked):
    import time
    time.sleep(0.05)            # (the caller leaves main() first)
    while parked:
        parked.pop()


def evaluate_predictions_on_lvis(lvis_gt, track_result, dt_columns, iou_type,
                                 logger):
    logger.info("Evaluating {} on LVIS...".format(track_result))
    from tao_amodal_amd.evaluation._core import timed
    with timed("image:results"):
        lvis_dt = LVISResults(lvis_gt, dt_columns)
    with timed("image:eval"):
        lvis_eval = LVISEval(lvis_gt, lvis_dt, iou_type)
        lvis_eval.run()
    lvis_eval.print_results()
    results = lvis_eval.get_results()
    results = {m: float(results[m] * 100) for m in LVIS_METRICS}
    logger.info("Evaluation results for {}: \n".format(iou_type)
                + create_small_table(results))
    logger.info("copypaste: " + ",".join(LVIS_METRICS))
    logger.info("copypaste: " + ",".join(
        "{0:.4f}".format(results[m]) for m in LVIS_METRICS))
    _PARKED.extend((lvis_dt, lvis_eval))
    return results


def eval_tao_track(ann_path, gt_dataset, gt_columns, dt_columns, logger):
    logger.setLevel(logging.INFO)
    results = {}
    logger.info("Loading gt {}...".format(ann_path))
    tao_gt = Tao(gt_dataset, columns=gt_columns)
    logger.info("Done")
    from tao_amodal_amd.evaluation._core import timed
    logger.info("Loading results...")
    with timed("track:unique_ids"):
        make_track_ids_unique(dt_columns)
    logger.info("Done")
    logger.info("Building")
    with timed("track:results"):
        tao_dt = TaoResults(tao_gt, dt_columns)
    with timed("track:eval"):
        tao_eval = TaoEval(tao_gt, tao_dt, logger=logger)
        logger.info("Done")
        tao_eval.run()
    tao_eval.print_results()
    res = tao_eval.get_results()
    results["TAO 3DmAP50"] = res["AP50"] * 100
    results["TAO 3DmAP50-HP"] = res["AP50-HP"] * 100
    results["TAO 3DmAP"] = res["AP"] * 100
    results["TAO 3DmAP-HP"] = res["AP-HP"] * 100
    keys = ["TAO 3DmAP50", "TAO 3DmAP50-HP", "TAO 3DmAP", "TAO 3DmAP-HP"]
    for k in keys:
        logger.info("{}:{:.4f}".format(k, results[k]))
    logger.info("copypaste: " + ",".join(keys))
    logger.info("copypaste: " + ",".join("{:.4f}".format(results[k]) for k in keys))
    _PARKED.extend((tao_gt, tao_dt, tao_eval))
    return results


class HeldLogs:
    """Holds back the log records of ONE thread on every handler in use and
    emits them later, in the order and on the handlers they were bound for."""

    class _Gate(logging.Filter):
        def __init__(self, owner, handler):
            super().__init__()
            self.owner, self.handler = owner, handler

        def filter(self, record):
            if self.owner.thread is not None and record.thread == self.owner.thread:
                self.owner.records.append((self.handler, record))
                return False
            return True

    def __init__(self, *loggers):
        import threading
        self._ident = threading.get_ident
        self.thread, self.records, self.gates = None, [], []
        seen = set()
        named = [logging.getLogger()] + list(loggers) + [
            lg for lg in logging.root.manager.loggerDict.values()
            if isinstance(lg, logging.Logger)]
        handlers = [h for lg in named for h in lg.handlers]
        if logging.lastResort is not None:
            handlers.append(logging.lastResort)
        for h in handlers:
            if id(h) not in seen:
                seen.add(id(h))
                gate = self._Gate(self, h)
                h.addFilter(gate)
                self.gates.append((h, gate))

    def run(self, fn, *args):
        self.thread = self._ident()
        return fn(*args)

    def close(self, replay):
        self.thread = None
        for h, gate in self.gates:
            h.removeFilter(gate)
        self.gates = []         # (owner <-> gates: no cycle left for the collector)
        if replay:
            for h, record in self.records:
                h.acquire()
                try:
                    h.emit(record)
                finally:
                    h.release()
        self.records = []


def main_distributed(args, annotation):
    """One rank of ``torchrun ... tools/eval_on_tao_amodal.py``."""
    import contextlib
    import io
    from tao_amodal_amd.columns import GTColumns
    from tao_amodal_amd.evaluation import _dist
    from tao_amodal_amd.evaluation._core import timed
    from tao_amodal_amd import dist as tdist, flatten_dev
    # stdout carries the result lines only: gloo / RCCL / the HIP runtime print
    # their banners to the C-level stdout, so descriptor 1 is pointed at stderr
    # for the run and Python's stdout at a copy of the real one
    sys.stdout.flush()
    saved_fd, saved_stdout = os.dup(1), sys.stdout
    py_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = py_stdout
    try:
        return _main_distributed(args, annotation)
    finally:
        # (a caller that runs the command in-process gets its stdout back)
        py_stdout.flush()
        sys.stdout = saved_stdout
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
        py_stdout.close()


def _main_distributed(args, annotation):
    import contextlib
    import io
    from tao_amodal_amd.columns import GTColumns
    from tao_amodal_amd.evaluation import _dist
    from tao_amodal_amd.evaluation._core import timed
    from tao_amodal_amd import dist as tdist, flatten_dev
    ctx = _dist.init_from_env()
    logger = logging.getLogger("__main__")
    logger.setLevel(logging.INFO if ctx.rank == 0 else logging.ERROR)
    handler = None
    if ctx.rank == 0:
        output_log = Path(args.output_log)
        output_log.parent.mkdir(parents=True, exist_ok=True)
        handler = logging.FileHandler(output_log, mode="w")
        logger.addHandler(handler)
    else:
        for name in ("tao.tao", "tao.results", "tao.eval", "root"):
            logging.getLogger(name).setLevel(logging.CRITICAL)
        logging.getLogger().setLevel(logging.CRITICAL)
    quiet = contextlib.redirect_stdout(io.StringIO()) if ctx.rank else contextlib.nullcontext()
    try:
        with quiet:
            with timed("parse"):
                gt = GTColumns.from_file_native(annotation)
                if gt is None:
                    with open(annotation) as f:
                        gt = GTColumns.from_json(json.load(f))
                dt = DTColumns.from_file_native(args.track_result, ctx.rank, ctx.world)
            with timed("exchange"):
                sh = _dist.shard_inputs(gt, dt, dt.first, ctx)
            # ---- image level
            lvis_gt = LVIS(annotation, columns=sh.gt_lvis)
            logger.info("Evaluating {} on LVIS...".format(args.track_result))
            lvis_eval = LVISEval(lvis_gt, LVISResults(lvis_gt, sh.dt_lvis, _share=True),
                                 "bbox", dist=ctx)
            lvis_eval.run()
            lvis_eval.print_results()
            results = lvis_eval.get_results()
            results = {m: float(results[m] * 100) for m in LVIS_METRICS}
            logger.info("Evaluation results for {}: \n".format("bbox")
                        + create_small_table(results))
            logger.info("copypaste: " + ",".join(LVIS_METRICS))
            logger.info("copypaste: " + ",".join(
                "{0:.4f}".format(results[m]) for m in LVIS_METRICS))
            # ---- track level
            logger.info("Loading gt {}...".format(annotation))
            tao_gt = Tao(annotation, columns=sh.gt_tao)
            logger.info("Done")
            logger.info("Loading results...")
            logger.info("Done")
            logger.info("Building")
            with timed("flatten"):
                # (a file with duplicate ids: every rank holds the whole set)
                universe = None if sh.whole else \
                    tdist.gather_visit_universe(sh.gt_tao, ctx.device, ctx.group)
                flat = flatten_dev.flatten_tao(sh.gt_tao, sh.dt_tao, device=ctx.device,
                                               visit_universe=universe)
            tao_eval = TaoEval(tao_gt, TaoResults(tao_gt, sh.dt_tao, _flat=flat, _share=True),
                               logger=logger, dist=ctx)
            logger.info("Done")
            tao_eval.run()
            tao_eval.print_results()
            res = tao_eval.get_results()
            out = {"TAO 3DmAP50": res["AP50"] * 100, "TAO 3DmAP50-HP": res["AP50-HP"] * 100,
                   "TAO 3DmAP": res["AP"] * 100, "TAO 3DmAP-HP": res["AP-HP"] * 100}
            for k, v in out.items():
                logger.info("{}:{:.4f}".format(k, v))
            logger.info("copypaste: " + ",".join(out))
            logger.info("copypaste: " + ",".join("{:.4f}".format(v) for v in out.values()))
    finally:
        if handler is not None:
            logger.removeHandler(handler)
            handler.close()
    sys.stdout.flush()
    import torch.distributed as dist
    dist.barrier(group=ctx.host_group)
    if os.environ.get("TAOAMD_TIMING") and ctx.rank == 0:
        from tao_amodal_amd.evaluation._core import TIMING
        print("taoamd timing (s): " + json.dumps(
            {k: round(v, 3) for k, v in TIMING.items()}), file=sys.stderr)


_EARLY = {}


def _early_hip_init():
    """A fresh process: the HIP runtime's own start-up (hsa_init, the primary
    context of the device: 0.2-0.4 s) on a helper thread WHILE torch is being
    imported -- native code that holds no interpreter lock.  The runtime is
    process-wide, so torch and the kernel library find the device initialised.
    The library loaded is the very file torch links (its bundled
    libamdhip64.so: one copy of the runtime in the process); anything unusual
    -- no such file, no device -- is left to the ordinary path."""
    import ctypes
    import importlib.util
    import time
    t = time.perf_counter()
    try:
        spec = importlib.util.find_spec("torch")
        so = os.path.join(spec.submodule_search_locations[0], "lib", "libamdhip64.so")
        if not os.path.exists(so):
            return
        hip = ctypes.CDLL(so, mode=ctypes.RTLD_GLOBAL)
        if hip.hipInit(0) != 0:
            return
        n = ctypes.c_int(0)
        if hip.hipGetDeviceCount(ctypes.byref(n)) != 0 or n.value < 1:
            return
        hip.hipSetDevice(0)
        hip.hipFree(None)               # creates the primary context
        _EARLY["hip_s"] = time.perf_counter() - t
        _EARLY["ok"] = True
    except Exception:
        pass
    finally:
        # (the prediction reader waits for this before it loads the kernel
        # library without torch: columns.DTColumns._from_file_device)
        from tao_amodal_amd import columns
        columns.EARLY_HIP["ok"] = bool(_EARLY.get("ok"))
        if columns.EARLY_HIP.get("event") is not None:
            columns.EARLY_HIP["event"].set()


def _warm_device(pred_path=None):
    """Everything a cold process pays once, on a helper thread beside the
    parse: torch's import, the HIP context, the kernel library and its code
    object (first launch), and device memory for the prediction columns and
    the tables in torch's caching allocator (a fresh process would otherwise
    meet ~50 hipMalloc calls between the parse and the first kernel)."""
    import time
    marks = [("start", time.perf_counter())]
    mark = lambda k: marks.append((k, time.perf_counter()))  # noqa: E731
    try:
        import torch
        mark("import torch")
        from tao_amodal_amd import _lib, flatten_dev  # noqa: F401
        mark("import package")
        if not torch.cuda.is_available():
            return
        mark("is_available")
        lib = _lib.load()
        mark("load library")
        box = torch.tensor([[0.0, 0.0, 1.0, 1.0]], dtype=torch.float64, device="cuda")
        out = torch.empty(1, dtype=torch.float64, device="cuda")
        mark("first tensors")
        lib.taoamd_bb_iou(box.data_ptr(), box.data_ptr(), 1, 1, None, out.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        mark("first launch")
        # torch's own device code is loaded op by op on first use: the handful
        # of primitives the table build uses (flatten_dev._cells_from_runs,
        # engine.DeviceProblem), once, on tiny tensors
        t = torch.arange(8, dtype=torch.int32, device="cuda")
        u = torch.unique(torch.cat([t, t]))
        i = torch.searchsorted(u, t)
        c = torch.cumsum(torch.bincount(i, minlength=8), 0)
        z = torch.zeros(9, dtype=torch.int64, device="cuda")
        z[i.long() + 1] = c
        w = torch.stack([t, t], dim=1).to(torch.int32).contiguous()
        (w[:, 0].clamp(max=3) << 2 | (t & 3)).long().cpu()
        torch.div(t, 2, rounding_mode="floor").index_select(0, t.long())
        torch.cuda.synchronize()
        mark("torch ops")
        if pred_path and os.path.exists(pred_path):
            free, _total = torch.cuda.mem_get_info()
            want = min(int(os.path.getsize(pred_path) * 1.5), int(free * 0.5))
            if want > (64 << 20):
                del box, out
                block = torch.empty(want, dtype=torch.uint8, device="cuda")
                del block           # (stays in the allocator's cache: split on demand)
        mark("allocation")
    except Exception:       # (whoever needs the device raises at the usual place)
        pass
    finally:
        if os.environ.get("TAOAMD_TIMING"):
            print("taoamd warm-up (s): " + ", ".join(
                "%s %.3f" % (k, t - marks[i][1]) for i, (k, t) in enumerate(marks[1:])),
                file=sys.stderr)


def main(argv=None):
    args = default_arg_parser(argv)
    annotation = args.annotation if args.annotation else DEFAULT_ANNOTATION
    # one rank of `torchrun ... eval_on_tao_amodal.py`: the launcher's own
    # variables (a WORLD_SIZE left in the environment by something else -- a
    # SLURM job, a notebook -- does not switch the mode: ADVICE r3)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "RANK" in os.environ \
            and "LOCAL_RANK" in os.environ and os.environ.get("TAOAMD_SINGLE", "0") == "0":
        return main_distributed(args, annotation)
    # ~1 s of a cold start -- importing torch, creating the HIP context,
    # loading the kernel library -- beside the parse instead of in front of the
    # first table (the readers are native code and the ground-truth halves of
    # the tables numpy: neither needs torch)
    import threading
    # (the main thread's numpy work -- hundreds of short calls -- must not wait
    # a full 5 ms GIL interval for another interpreter thread after each of them)
    sys.setswitchinterval(2e-4)
    # A fresh process imports torch ON THIS THREAD while both readers -- native
    # code, no interpreter lock held -- run in the background; the rest of the
    # warm-up (context, library, allocator) then goes to a helper beside the
    # ground-truth halves of the tables.  (Round 4: with the import on the
    # helper, beside this thread's numpy calls, the two took turns with the
    # interpreter lock and 0.75 s of import became 1.15-1.2.)
    cold = "torch" not in sys.modules
    if not cold:
        threading.Thread(target=_warm_device, args=(args.track_result,), daemon=True).start()
    output_log = Path(args.output_log)
    logger = logging.getLogger("__main__")
    logger.setLevel(logging.INFO)
    output_log.parent.mkdir(parents=True, exist_ok=True)
    handler = logging.FileHandler(output_log, mode="w")
    logger.addHandler(handler)
    from tao_amodal_amd.evaluation._core import TIMING, timed
    pool = lvis_gt = dt_columns = dt_future = gt_future = track = held = None
    try:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=2 if cold else 1)
        with timed("parse"):
            # the native readers run outside the GIL: the two files are
            # parsed side by side
            def capped(team, fn, *a):
                """fn(*a) with the host library's teams of the calling thread
                held to `team` threads: beside the import -- one serial thread --
                two teams of quota-many threads use the control group's CPU
                quota up in half of every scheduler period and all are frozen
                for the rest of it, the import included."""
                from tao_amodal_amd import flatten as fl
                lib = fl._host_lib() if team else False
                if not lib or not hasattr(lib, "taoamd_host_thread_cap"):
                    return fn(*a)
                lib.taoamd_host_thread_cap(team)
                try:
                    return fn(*a)
                finally:
                    lib.taoamd_host_thread_cap(0)
            teams = (0, 0)
            if cold:
                from tao_amodal_amd import flatten as fl
                lib = fl._host_lib()
                q = lib.taoamd_host_threads() - 1 if lib else 0
                early = not os.environ.get("TAOAMD_NO_EARLY_HIP")
                if q >= 7:
                    teams = (q - q // 3, q // 3)        # (predictions, annotations)
                    if early and os.environ.get("TAOAMD_DEVICE_INGEST", "1") != "0":
                        # the prediction file is read on the device once the HIP
                        # runtime is up (started below, beside the import): the
                        # cores go to the annotation reader
                        teams = (q // 3, q - q // 3)
                if early:
                    # (before the readers start: the prediction reader waits for
                    # the runtime, then loads the kernel library without torch)
                    from tao_amodal_amd import columns as _columns
                    _columns.EARLY_HIP["event"] = threading.Event()
                    threading.Thread(target=_early_hip_init, daemon=True).start()
            dt_future = pool.submit(capped, teams[0], DTColumns.from_json, args.track_result)

            def read_annotation():
                gt = LVIS(annotation)           # native reader when built
                gt.columns
                return gt
            if cold:
                gt_future = pool.submit(capped, teams[1], read_annotation)
                with timed("parse:import_torch"):
                    import torch  # noqa: F401
                from tao_amodal_amd import columns as _columns
                _columns.EARLY_HIP["torch_loaded"] = True
                threading.Thread(target=_warm_device, args=(args.track_result,),
                                 daemon=True).start()
                with timed("parse:annotation"):
                    lvis_gt = gt_future.result()
            else:
                with timed("parse:annotation"):
                    lvis_gt = read_annotation()
            gt_dataset = annotation          # the track level shares the columns
            if not cold:
                # a running process: the halves of the cell tables that depend on
                # the annotation file alone start on threads of their own now;
                # each level's table build picks its half up (or waits for it)
                from tao_amodal_amd import prepare
                prepare.prepare_gt(lvis_gt.columns, wait=False)
            elif cold or not dt_future.done():
                # the annotation file is the smaller one: its halves of the
                # cell tables are built while the predictions are still read
                # (a fresh process: while the helper creates the HIP context.
                # Round 6 measured them on the worker instead, beside the import
                # of torch: the import then takes 1.0 instead of 0.9 s and the
                # levels start before torch's device code is loaded -- 1.9-2.3 s
                # either way)
                with timed("parse:gt_halves"):
                    from tao_amodal_amd import prepare
                    prepare.prepare_gt(lvis_gt.columns)
            with timed("parse:wait_predictions"):
                dt_columns = dt_future.result()
        if len(dt_columns) and not os.environ.get("TAOAMD_CLI_SERIAL"):
            # the track level runs beside the image level on the worker
            # thread (most of either is numpy or the GPU: no GIL held); what
            # it logs is held back and written once the image level is done,
            # so every stream and the log file read as if one followed the other
            held = HeldLogs(logger)
            track = pool.submit(held.run, eval_tao_track, annotation, gt_dataset,
                                lvis_gt.columns, dt_columns, logger)
            try:
                # (round 6: each level on a HIP stream of its own -- so that what
                # one reads back waits for its own kernels only -- measured no
                # faster, 0.92-1.17 s either way: the GPU is busy for 0.1 s of it)
                evaluate_predictions_on_lvis(lvis_gt, args.track_result, dt_columns,
                                             "bbox", logger)
            except BaseException:
                track.exception()       # (wait: nothing of it is shown)
                held.close(replay=False)
                raise
            failed = track.exception()
            held.close(replay=True)
            if failed is not None:
                raise failed
        else:
            evaluate_predictions_on_lvis(lvis_gt, args.track_result, dt_columns,
                                         "bbox", logger)
            eval_tao_track(annotation, gt_dataset, lvis_gt.columns, dt_columns,
                           logger)
    finally:
        if pool is not None:
            pool.shutdown(wait=False)
        logger.removeHandler(handler)
        handler.close()
        # (every name of this frame that holds a table goes with them)
        parked = _PARKED[:]
        del _PARKED[:]
        parked.extend((lvis_gt, dt_columns, dt_future, gt_future, track, held))
        lvis_gt = dt_columns = dt_future = gt_future = track = held = None
        threading.Thread(target=_let_go, args=(parked,), daemon=True).start()
        del parked
    if os.environ.get("TAOAMD_TIMING"):
        # wall-clock split (stderr only: stdout and the log file stay identical
        # to the reference's)
        if "hip_s" in _EARLY:
            TIMING["parse:early_hip_init"] = _EARLY["hip_s"]
        print("taoamd timing (s): " + json.dumps(
            {k: round(v, 3) for k, v in TIMING.items()}), file=sys.stderr)


if __name__ == "__main__":
    main()
    # The tables are printed and the log file is closed: leave without the
    # interpreter's and the HIP runtime's tear-down (0.3-0.4 s of a fresh
    # process: module finalisers, the context, gigabytes of arrays freed page by
    # page).  Exit status 0 like the reference's script; an exception above
    # takes the ordinary way out with its traceback.
    # (one rank of a torchrun job leaves the ordinary way: its process group
    # and RCCL's communicators are torn down in order)
    if not os.environ.get("TAOAMD_SLOW_EXIT") and \
            int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        logging.shutdown()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
