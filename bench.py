#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X evaluation hot path.

    python bench.py --gpus N --steps K --warmup W [--config 3s|2|5s] [--decimal]
                    [--scaling weak|strong]

One "step" = one pass of the hot path over one batch of synthetic input that
is already resident in HBM: for BOTH evaluators (image-level LVISEval and
track-level TaoEval) range masks -> (category, -score) sort -> [3D track IoU]
-> IoU + greedy match at 10 thresholds x {6 | 20} ranges -> accumulate
(precision[T,R,K,A] and recall materialised in the reference layout).

Workloads (BASELINE.json configs):
    --config 3s  (default) the largest single-GPU configuration: the stand-in
                 for the full validation set (its JSONs are not in the
                 container) at the size north_star names -- 2000 videos x 300
                 frames x 50 dets, 1203 categories, 30 M boxes in
    --config 2   "Synthetic 200 videos x 300 frames x 50 dets" (the bit-exact
                 parity configuration)
    --config 5s  stand-in for the stress set on ONE GPU: 10 000 videos x 1 frame
                 x 1000 dets (10 M boxes; the top-300 cut per image is part of
                 the host flatten time reported in host_s)

N > 1 (one process per GPU, RCCL): launched by the driver's torchrun, or --
when WORLD_SIZE is not set -- by this script itself, which spawns N ranks and
fails loudly if the box has fewer GPUs.  `--scaling weak` (default): every rank
holds one whole shard of the configuration, the data set grows with N.
`--scaling strong`: ONE fixed set of the configuration split by video over the
ranks (BASELINE.json Config 4).  `--shard unit` (default, the partition
BASELINE.json names): a rank keeps its own videos, the per-detection records
travel to the category owners in one all_to_all, are merged run by run and
swept, and the result tables are assembled with one run-length packed
all_gather per evaluator.  `--shard category`: every rank evaluates a
contiguous category block of the union of all shards (no record exchange, but
every rank flattens the whole input) (tao_amodal_amd/dist.py).

Prints ONE JSON line on rank 0 (see the driver contract in the task text).
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

# (before numpy / torch / the native libraries load libgomp: tao_amodal_amd/__init__.py)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec
MIN_TIMED_S = 0.25        # floor of the timed region, whatever --steps says
PROBE_EVERY = 8           # steps between two per-kernel event probes

CONFIGS = {
    "2": dict(videos=200, frames=300, dets=50,
              name="SYNTH Config 2"),
    "3s": dict(videos=2000, frames=300, dets=50,
               name="SYNTH Config 3 stand-in (full-validation scale)"),
    "5s": dict(videos=10000, frames=1, dets=1000,
               name="SYNTH Config 5 stand-in (stress, top-300 cut per image)"),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=None)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--config", choices=sorted(CONFIGS), default="3s")
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                   help="N > 1: a shard of the configuration per rank (weak) or "
                        "one fixed set split by video over the ranks (strong)")
    p.add_argument("--no-wallclock", action="store_true",
                   help="skip the end-to-end leg (the set written as JSON files, "
                        "the drop-in CLI run on them)")
    p.add_argument("--videos", type=int, default=None)
    p.add_argument("--frames", type=int, default=None)
    p.add_argument("--dets", type=int, default=None)
    p.add_argument("--cats", type=int, default=1203)
    p.add_argument("--seed", type=int, default=20240807)
    p.add_argument("--cpu-sample-videos", type=int, default=2000,
                   help="videos of the same workload timed through the C "
                        "oracle for cpu_baseline (rank 0, N=1 only); the whole "
                        "set when it has no more than that")
    p.add_argument("--decimal", action="store_true",
                   help="decimal box coordinates, as real prediction files have: "
                        "the track level runs with the frame-order guard active")
    p.add_argument("--no-decimal-leg", action="store_true",
                   help="skip the second measurement of the default invocation: "
                        "the same workload with decimal coordinates (frame-order "
                        "guard active), 20 steps, in a subprocess")
    p.add_argument("--leg", choices=["decimal"], default=None,
                   help="internal: this process is the decimal leg of another "
                        "bench run (implies --decimal --no-wallclock; the oracle "
                        "verifies on all host cores, no cpu_baseline timing)")
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--serial", action="store_true",
                   help="run the two evaluator passes back to back on one stream")
    p.add_argument("--shard", choices=["category", "unit"], default="unit",
                   help="multi-GPU partition: the ranks' own videos (default; "
                        "records meet at the category owners) or contiguous "
                        "category blocks of one shared input (no record "
                        "exchange, but every rank reads the whole input)")
    p.add_argument("--emulate", default=None, metavar="W:R",
                   help="with --force-dist on one GPU: run the per-rank work of "
                        "rank R of a W-rank category-sharded job (diagnostic; "
                        "the printed value counts this rank's pairs only)")
    p.add_argument("--force-dist", action="store_true",
                   help="take the multi-GPU code path even with one rank")
    a = p.parse_args()
    if a.leg == "decimal":
        a.decimal, a.no_wallclock, a.no_decimal_leg = True, True, True
    cfg = CONFIGS[a.config]
    for k in ("videos", "frames", "dets"):
        if getattr(a, k) is None:
            setattr(a, k, cfg[k])
    return a


# --------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torchrun around it
# --------------------------------------------------------------------------
def self_spawn(n):
    """Start n ranks of this very command (one per GPU) and wait for them."""
    import torch
    have = torch.cuda.device_count()
    if have < n:
        sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on "
                         "this box -- refusing to run fewer ranks than asked\n"
                         % (n, have))
        sys.exit(2)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable] + sys.argv, env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    if rc:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


# --------------------------------------------------------------------------
# algorithmic (compulsory) HBM bytes of ONE launch of every kernel of the
# step, DESIGN.md "Kernels": each datum the kernel needs moved once
# --------------------------------------------------------------------------
def _lib_tile():
    from tao_amodal_amd import _lib
    return _lib.SEGMENT_TILE


def kernel_models(dp, ws):
    T, R = 10, 101
    K, A, nw = dp.n_cat, dp.n_rng, dp.n_words
    n_dt, n_gt, n_cells = dp.n_dt, dp.n_gt, dp.n_cells
    rows = 16 * nw * n_dt                          # matched + ignored words
    live = int((ws.num_gt > 0).sum().item())       # (category, range) rows with GT
    table = 8 * T * R                              # one row of val / precision
    seg = np.diff(dp.cat_off_host).astype(np.int64)
    tile = _lib_tile()
    n_multi = int(seg[seg > tile].sum())
    m = {}
    if dp.kind == "lvis":
        m["lvis_ranges_kernel"] = n_gt * (8 + 1 + 4 + 4) + n_dt * (1 + 4)
        m["match_group_kernel"] = (n_dt * (32 + 4 + 1 + 4 + 4 + 16 * nw)
                                   + n_gt * (32 + 4 + 1) + n_cells * 8)
    else:
        m["tao_ranges_kernel"] = n_gt * (8 + 4 + 4 + 1 + 4 + 4) + n_dt * (8 + 4 + 1 + 4)
        m["match_group_kernel"] = (n_dt * (4 + 1 + 4 + 4 + 16 * nw) + dp.n_iou * 8
                                   + n_gt * (4 + 1) + n_cells * 8)
        frames = dp.t["dt_frame_pos"].numel() + dp.t["gt_frame_pos"].numel()
        if dp.t.get("tasks") is not None:
            m["track_iou_task_kernel"] = (
                frames * 32 + dp.n_iou * 8 + dp.n_tasks * 16
                + dp.t["task_rows"].numel() * (4 + 16)
                + dp.t["task_pairs"].numel() * (4 + 8))
        m["track_iou_kernel"] = frames * 36 + (n_dt + n_gt) * 4 + dp.n_iou * 8
    m["count_gt_kernel"] = n_gt * 4 + K * A * 4
    m["seg_tile_kernel"] = n_dt * (8 + 8)           # score in; order + dst (or key + index) out
    m["seg_kmerge_kernel"] = n_multi * (12 + 8)     # key + index in; order + dst out
    m["seg_mpass_kernel"] = n_multi * 24
    m["seg_bucket_kernel"] = n_multi * (12 + 8)     # key + index in; order + dst out
    m["seg_split_kernel"] = n_multi // 16 * 12      # every 16th (key, index) as a sample
    # sample sort (taoamd_sort_sampled): samples -> splitters; one scatter pass
    # (score in, key + index into the bucket's slots); one register sort per
    # bucket (key + index in, order + dst out)
    nc, ns_, nt_, nb_ = getattr(dp, "ss_sizes", (0, 0, 0, 0))
    n_ss = int(seg[seg > 1024].sum())               # elements of the split chunks
    m["ss_split_kernel"] = nb_ * (32 * 8 + 12)      # ~32 samples per bucket in, a splitter out
    m["ss_scatter_kernel"] = n_ss * (8 + 12) + nb_ * 12
    m["ss_sort_kernel"] = n_dt * (12 + 8)
    m["acc_sweep_kernel"] = rows + live * table     # one-pass sweep: rows once, records out
    m["acc_sccount_kernel"] = rows
    m["acc_raise_kernel"] = 2 * live * table
    m["acc_count_kernel"] = rows
    m["acc_chunkmax_kernel"] = rows
    m["acc_emit_kernel"] = rows + live * table
    m["acc_fused_kernel"] = rows + live * table
    m["acc_finalize_kernel"] = live * table + K * A * (table + 8 * T)
    fused = dp.kind == "lvis" and not dp.mask_iou
    variants = {"match_group_kernel": "<true>" if fused else "<false>",
                "match_kernel": "<true>" if fused else "<false>",
                "match_big_kernel": "<true>" if fused else "<false>"}
    if dp.kind == "tao":
        # the timed passes launch the instance that does not count the pairs'
        # common frames (a workspace's first pass does: engine.stage_track_iou)
        variants["track_iou_task_kernel"] = "<%d, %s>" % (
            dp.iou_mode, "false" if dp.iou_mode == 0 else "true")
    grids = {"seg_tile_kernel": dp.n_tiles * 256,
             "match_group_kernel": (dp.n_groups + 3) // 4 * 256,
             "lvis_ranges_kernel": (max(n_gt, n_dt) + 255) // 256 * 256,
             "tao_ranges_kernel": (max(n_gt, n_dt) + 255) // 256 * 256,
             "seg_mpass_kernel": dp.n_tiles * 256,
             "seg_bucket_kernel": dp.n_tiles * 2 * 256,
             "seg_split_kernel": K * 256,
             "seg_kmerge_kernel": (n_dt + 255) // 256 * 256,
             "acc_finalize_kernel": ((K * A + 63) // 64) * ((T * R + 63) // 64) * 256,
             # (eight runs cut at category boundaries: accumulate.hip, XCD-aware order)
             "acc_sweep_kernel": 8 * (((n_dt // 2048) + K + 1) // 8
                                      + int(seg.max() if len(seg) else 0) // 2048 + 3) * nw * 256,
             "acc_raise_kernel": (K * A * T + 3) // 4 * 256,
             "acc_cj_kernel": (K * A * R + 255) // 256 * 256,
             "ss_scatter_kernel": nt_ * 256,
             "ss_sort_kernel": (nb_ + 3) // 4 * 256,
             "ss_split_kernel": (ns_ + 3) // 4 * 256}
    return m, grids, variants


def survey_models(dp, ws):
    """SURVEY.md 8(d)'s algorithmic bytes per launch, strictly: every datum of
    the reference's interface moved once -- boxes 32 B, score 8 B, range /
    ignore masks 1 B (image level) or 4 B (track level), 16 B per cell, the
    packed (TP, ignore) rows, the fixed output tables -- and NOTHING of the
    build's own indirections (dst[], dt_meta, the launch plans, the sort's
    slots).  The sort stage's compulsory traffic is the scores, read once
    (counted on the scatter pass); kernels that move intermediates only have
    no compulsory byte and are left out."""
    T, R = 10, 101
    K, A, nw = dp.n_cat, dp.n_rng, dp.n_words
    n_dt, n_gt, n_cells = dp.n_dt, dp.n_gt, dp.n_cells
    rows = 16 * nw * n_dt
    live = int((ws.num_gt > 0).sum().item())
    table = 8 * T * R
    m = {}
    if dp.kind == "lvis":
        m["match_group_kernel"] = n_dt * (32 + 1 + 16 * nw) + n_gt * (32 + 1) + n_cells * 16
        m["lvis_ranges_kernel"] = n_gt * (8 + 1 + 1) + n_dt * (1 + 1)
    else:
        m["match_group_kernel"] = (dp.n_iou * 8 + n_dt * (4 + 16 * nw) + n_gt * 4
                                   + n_cells * 16)
        frames = dp.t["dt_frame_pos"].numel() + dp.t["gt_frame_pos"].numel()
        m["track_iou_task_kernel"] = frames * 32 + dp.n_iou * 8
        m["track_iou_kernel"] = frames * 32 + dp.n_iou * 8
        m["tao_ranges_kernel"] = n_gt * (8 + 4 + 4 + 4) + n_dt * (8 + 4 + 4)
    m["ss_scatter_kernel"] = n_dt * 8
    m["seg_tile_kernel"] = n_dt * 8
    for k in ("acc_sweep_kernel", "acc_emit_kernel", "acc_fused_kernel"):
        m[k] = rows + live * table
    m["acc_finalize_kernel"] = K * A * (table + 8 * T)          # the output tensors, once
    return m


def step_algorithmic_bytes(dpl, dpt):
    """B_alg of one step (both evaluators)."""
    frames = dpt.t["dt_frame_pos"].numel() + dpt.t["gt_frame_pos"].numel()
    fixed = 8 * 10 * (101 + 1) * dpl.n_cat * (dpl.n_rng + dpt.n_rng)
    return (32 * (dpl.n_dt + dpl.n_gt) + 32 * frames + (8 + 1 + 16) * dpl.n_dt
            + (8 + 4 + 64) * dpt.n_dt + 17 * (dpl.n_cells + dpt.n_cells) + fixed)


def sources_digest():
    """Digest of the kernel sources (tao_amodal_amd/csrc): a PMC summary speaks
    for the kernels it was taken on, tools/prof_summary.py records the digest
    and a summary of other sources is not quoted."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "tao_amodal_amd", "csrc", "*"))):
        if path.endswith((".hip", ".hpp", ".cpp", ".sh")):
            h.update(os.path.basename(path).encode())
            with open(path, "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, variant, grid, workload):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary
    of THIS workload (profiles/*_pmc.json written by tools/prof_summary.py
    from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very
    command and of these very kernel sources); None when no summary of the
    workload is present or the launch cannot be told from the other evaluator's.  `variant`: the template
    arguments of the instance this pass launches ("<true>"), `grid`: its grid
    size in work items -- the image level and the track level launch the same
    kernels, the summary keys them by full name and grid."""
    import glob
    import re

    def version(path):      # r02_v3_pmc.json -> (2, 3): numeric, not lexical
        mm = re.search(r"r(\d+)_v(\d+)", os.path.basename(path))
        return (int(mm.group(1)), int(mm.group(2))) if mm else (-1, -1)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")),
                       key=version, reverse=True):
        with open(path) as f:
            d = json.load(f)
        if d.get("workload") != workload:
            continue
        if d.get("sources") != sources_digest():
            # taken on other kernels than the ones that run now (VERDICT r3:
            # the line quoted a 36-row kernel's traffic for the 32-row one)
            return {"bytes": None, "stale": os.path.basename(path)}
        ents = []
        for name, ee in d["kernels"].items():
            mm = re.match(r"(?:void )?%s(<[^(]*>)?\(" % re.escape(kernel), name)
            # (the variant names the FIRST template argument: "<true>" is
            # match_group_kernel<true, ...>, whatever switches follow it)
            if not mm or (variant and not (mm.group(1) or "").startswith(variant.rstrip(">"))):
                continue
            ents += [e for e in ee if e.get("hbm_bytes_corrected") is not None]
        if grid is not None:
            # the launch of THIS pass only: an entry of another grid is the
            # other evaluator's (or another problem's) launch of the kernel
            ents = [e for e in ents if e.get("grid") == grid]
        if len(ents) == 1:
            return {"bytes": ents[0]["hbm_bytes_corrected"],
                    "source": os.path.basename(path)}
        return None
    return None


def hbm_delivers():
    """What HBM was MEASURED to deliver on an MI355X (tools/hbm_rates.hip, the
    newest profiles/*_hbm_rates.txt): context for the 8 TB/s the roofline
    fractions are taken against, never their denominator."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_rates.txt")))
    if not files:
        return None
    best = {}
    for line in open(files[-1]):
        m = re.match(r"(read|write|copy)\s.*\s([0-9.]+) TB/s", line)
        if m:
            best[m.group(1)] = max(best.get(m.group(1), 0.0), float(m.group(2)) * 1000.0)
    if not best:
        return None
    return dict({k + "_GBs": v for k, v in best.items()},
                source=os.path.relpath(files[-1], ROOT) + " (tools/hbm_rates.hip: best of "
                "1-8 KB pieces in sequence / shuffled, 2 GiB working sets)")


def wallclock_leg(gt, dt):
    """tools/eval_on_tao_amodal.py (the plugin surface) on the workload written
    out as prediction.json / annotation JSON: total seconds and the parse /
    flatten / upload+plan / kernels / download / summarize split."""
    import contextlib
    import importlib.util
    import io
    import shutil
    import tempfile
    import threading
    d = tempfile.mkdtemp(prefix="taoamd_wall_", dir="/tmp")
    try:
        gt_p, pr_p = os.path.join(d, "gt.json"), os.path.join(d, "pred.json")
        t0 = time.perf_counter()
        gt.write_json(gt_p)
        dt.write_json(pr_p)
        t_write = time.perf_counter() - t0
        # (4 GB of freshly written pages: their write-back to the disk is waited
        # for here, not left to run beside the timed calls -- a user's files
        # are clean pages of the page cache too)
        os.sync()
        sizes = {"gt_MB": round(os.path.getsize(gt_p) / 1e6, 1),
                 "pred_MB": round(os.path.getsize(pr_p) / 1e6, 1)}
        os.environ["TAOAMD_TIMING"] = "1"
        spec = importlib.util.spec_from_file_location(
            "taoamd_cli", os.path.join(ROOT, "tools", "eval_on_tao_amodal.py"))
        cli = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(cli)
        from tao_amodal_amd.evaluation._core import TIMING
        TIMING.clear()
        text = io.StringIO()
        # (what the bench's earlier legs left behind -- the oracle's tables, the
        # writers' buffers: gigabytes in reference cycles -- is collected HERE:
        # the interpreter's collector would otherwise run, and unmap them, in
        # the middle of the timed call: 0.55 s became 0.75-1.4 s, measured)
        import gc
        gc.collect()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(text), contextlib.redirect_stderr(io.StringIO()):
            cli.main(["--track_result", pr_p, "--annotation", gt_p, "--output_log",
                      os.path.join(d, "eval.log")])
        total = time.perf_counter() - t0
        lines = text.getvalue().splitlines()
        # the same command as a user runs it: a fresh interpreter (imports, HIP
        # context, loading the libraries) -- the files are in the page cache.
        # (The call above leaves helper threads behind -- the columns' host
        # copies, the reader's release -- and its own garbage: both are let go
        # first, a fresh process competes with neither.)
        for th in threading.enumerate():
            if th is not threading.current_thread() and \
                    not th.name.startswith(("ThreadPoolExecutor", "pydev")):   # (idle pool workers stay)
                th.join(timeout=2.0)
        gc.collect()
        t0 = time.perf_counter()
        r = subprocess.run(
            [sys.executable, os.path.join(ROOT, "tools", "eval_on_tao_amodal.py"),
             "--track_result", pr_p, "--annotation", gt_p, "--output_log",
             os.path.join(d, "eval_cold.log")],
            env=dict(os.environ, TAOAMD_TIMING="1"), capture_output=True, text=True)
        cold = {"total": round(time.perf_counter() - t0, 3), "rc": r.returncode,
                "same_stdout": r.stdout == text.getvalue(),
                "what": "python tools/eval_on_tao_amodal.py as a subprocess on the "
                        "same files: interpreter start, imports, HIP context, "
                        "library load included; files in the page cache"}
        for line in r.stderr.splitlines():
            if line.startswith("taoamd timing (s): "):
                cold["split"] = json.loads(line[len("taoamd timing (s): "):])
        return {"total": round(total, 3), "cold": cold,
                "split": {k: round(v, 3) for k, v in TIMING.items()},
                "files": sizes, "write_files_s_not_counted": round(t_write, 2),
                "what": "tools/eval_on_tao_amodal.py in this process on the same "
                        "workload as JSON files: both evaluators, printed tables "
                        "(gc.collect() of the bench's own earlier garbage first)",
                "first_line": lines[0] if lines else None}
    except Exception as e:      # (a full /tmp must not lose the bench line)
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        os.environ.pop("TAOAMD_TIMING", None)
        shutil.rmtree(d, ignore_errors=True)


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and (args.gpus or 1) > 1:
        self_spawn(args.gpus)
    world = int(world_env or "1")
    if args.gpus is not None and args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d "
                         "rank(s)\n" % (args.gpus, world))
        sys.exit(2)
    # stdout carries exactly one JSON line: RCCL and the HIP runtime print
    # banners to the C-level stdout, so fd 1 is pointed at stderr for the run
    # and the line goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if world > 1 or args.force_dist:
        # the category-partitioned step keeps 4 compute streams + the RCCL
        # stream busy; with the default of 4 hardware queues per process they
        # alias and serialise (measured 1.25 -> 0.94 ms/step with 8)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if torch.cuda.device_count() <= local:
        sys.stderr.write("bench.py: rank %d needs GPU %d, %d visible\n"
                         % (rank, local, torch.cuda.device_count()))
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ranks_verified = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=dev)
        ones = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(ones)
        ranks_verified = int(ones.item())
        assert dist.get_world_size() == world and ranks_verified == world, \
            "RCCL group has %d ranks, %d asked for" % (ranks_verified, world)

    from tao_amodal_amd import _lib, engine, flatten
    from tao_amodal_amd.synth import synth

    by_category = use_dist and args.shard == "category"
    data_world, data_rank = world, rank
    if args.emulate:
        assert by_category and world == 1, "--emulate needs --force-dist on one GPU"
        data_world, data_rank = (int(x) for x in args.emulate.split(":"))
    strong = use_dist and args.scaling == "strong"
    t0 = time.time()
    if strong:
        # ONE fixed set (the configuration as named), split by video: every
        # rank generates it from the one seed and keeps a contiguous block of
        # videos (by-video partition) or the whole set (by-category partition,
        # where a rank evaluates its category block of everything)
        gt, dt = synth(seed=args.seed, V=args.videos, F=args.frames, C=args.cats,
                       dets_per_frame=args.dets, decimal=args.decimal)
        if not by_category:
            lo = args.videos * data_rank // data_world
            hi = args.videos * (data_rank + 1) // data_world
            keep = np.zeros(args.videos, dtype=bool)
            keep[lo:hi] = True            # (synth: vid_id ascending)
            mine = gt.vid_id[keep]
            dt = dt.take(np.flatnonzero((dt.video_id >= mine[0])
                                        & (dt.video_id <= mine[-1])))
            gt = gt.select_videos(keep)
    elif by_category:
        # weak scaling: the data set grows with the number of ranks (one shard
        # of the configuration per rank) and every rank evaluates its category
        # block of the WHOLE set, so the work per GPU stays fixed
        from tao_amodal_amd.columns import DTColumns, GTColumns
        parts = [synth(seed=args.seed + r, V=args.videos, F=args.frames,
                       C=args.cats, dets_per_frame=args.dets,
                       video_id_base=r * args.videos, decimal=args.decimal)
                 for r in range(data_world)]
        gt = GTColumns.concat([p[0] for p in parts])
        dt = DTColumns.concat([p[1] for p in parts])
        del parts
    else:
        gt, dt = synth(seed=args.seed + rank, V=args.videos, F=args.frames,
                       C=args.cats, dets_per_frame=args.dets,
                       video_id_base=rank * args.videos, decimal=args.decimal)
    t_gen = time.time() - t0
    n_boxes_in = len(dt)
    t0 = time.time()
    if by_category:
        # (the shards are cut out of host tables: flatten.py)
        fl = flatten.flatten_lvis(gt, dt)
        dt.track_id, _ = flatten.make_track_ids_unique(dt)
        ft = flatten.flatten_tao(gt, dt)
    elif use_dist:
        # a rank's own videos, tables built on the device; the visiting order
        # comes from the image ids of ALL ranks (one all_gather)
        from tao_amodal_amd import dist as tdist, flatten_dev
        universe = tdist.gather_visit_universe(gt, dev)
        fl = flatten_dev.flatten_lvis(gt, dt, device=dev)
        dt.track_id, _ = flatten.make_track_ids_unique(dt)
        ft = flatten_dev.flatten_tao(gt, dt, device=dev, visit_universe=universe)
        torch.cuda.synchronize()
    else:
        # cell tables built on the device (flatten_dev / csrc/flatten.hip)
        from tao_amodal_amd import flatten_dev
        fl = flatten_dev.flatten_lvis(gt, dt, device=dev)
        dt.track_id, _ = flatten.make_track_ids_unique(dt)
        ft = flatten_dev.flatten_tao(gt, dt, device=dev)
        torch.cuda.synchronize()
    if by_category:
        from tao_amodal_amd import dist as tdist
        k0, k1, _ = tdist.category_block(len(fl.cat_ids), data_rank, data_world)
        fl = tdist.shard_by_category(fl, k0, k1)
        ft = tdist.shard_by_category(ft, k0, k1)
    t_flat = time.time() - t0
    t0 = time.time()
    dpl, dpt = engine.DeviceProblem(fl, dev), engine.DeviceProblem(ft, dev)
    wsl, wst = engine.Workspace(dpl), engine.Workspace(dpt)
    torch.cuda.synchronize()
    t_h2d = time.time() - t0

    if use_dist:
        from tao_amodal_amd import dist as tdist
        if by_category:
            plan = tdist.CategoryPlan(dpl, dpt, rank, world, dev)
        else:
            plan = tdist.ExchangePlan(dpl, dpt, rank, world, dev)

        def step():
            plan.step()
    elif args.serial:
        def step():
            engine.run(dpl, wsl)
            engine.run(dpt, wst)
    else:
        # the two evaluator passes are independent: overlap them on HIP streams
        overlap = engine.Overlap(dev)

        def step():
            overlap.run_pair(dpl, wsl, dpt, wst)

    def bracket():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # ---- how many repetitions of the K steps make the timed region >= 0.25 s
    bracket()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    bracket()
    calib = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(calib, op=dist.ReduceOp.MAX)
    reps = max(1, int(math.ceil(MIN_TIMED_S / max(float(calib.item()), 1e-6))))
    timed_steps = reps * args.steps

    # ---- the timed region: reps x K steps between two barrier + synchronize
    _lib.kernel_timings()                  # forget anything recorded so far
    # (look-backs of the one-pass sweep that give up anywhere in the region
    # count into this word; the workspaces' own flags are per pass -- an
    # unprepared pass, every pass of the multi-GPU plans, clears its own)
    giveups = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().taoamd_accumulate_giveup_counter(giveups.data_ptr()),
               "taoamd_accumulate_giveup_counter")
    bracket()
    t0 = time.perf_counter()
    for i in range(timed_steps):
        # (every kernel of every PROBE_EVERY-th step is bracketed with events
        # on its own stream: two event records per launch cost host time)
        probing = rank == 0 and i % PROBE_EVERY == PROBE_EVERY // 2
        if probing:
            _lib.kernel_timing(True)
        step()
        if probing:
            _lib.kernel_timing(False)
    host_ms = (time.perf_counter() - t0) / timed_steps * 1e3   # launch side only
    bracket()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    in_step = _lib.kernel_timings() if rank == 0 else {}
    # ---- the one-pass sweep's look-back: waits that gave up anywhere in the
    # timed region are counted in `giveups` (every pass, every plan); the flag
    # of the workspace that was swept tells of the LAST pass (of all prepared
    # passes on one GPU: nothing clears it there) -- that pass is then swept
    # again with the chunked kernels (what the CLI does) and the line says so
    _lib.check(_lib.load().taoamd_accumulate_giveup_counter(None),
               "taoamd_accumulate_giveup_counter")
    look_back_giveups = int(giveups.item())
    if use_dist:
        gsum = torch.tensor([look_back_giveups], dtype=torch.int64, device=dev)
        dist.all_reduce(gsum)
        look_back_giveups = int(gsum.item())
    if use_dist:
        plan.lvis.check()
        plan.tao.check()
        look_back_timeouts = int(getattr(plan.lvis, "sweep_recovered", 0) +
                                 getattr(plan.tao, "sweep_recovered", 0))
    else:
        look_back_timeouts = int(engine.sweep_ok(dpl, wsl)) + int(engine.sweep_ok(dpt, wst))
    # ---- every kernel alone: a few serial steps, one stream, events on
    alone = {}
    if rank == 0 and not use_dist:
        _lib.kernel_timing(True)
        for _ in range(5):
            engine.run(dpl, wsl)
            engine.run(dpt, wst)
        torch.cuda.synchronize()
        _lib.kernel_timing(False)
        alone = _lib.kernel_timings()

    # ---- pairs: exact counts from the cell tables / the kernel's counter
    p_l = dpl.n_pairs
    p_t = int(plan.pair_frames()) if use_dist else int(wst.pair_frames.item())
    pairs = torch.tensor([p_l + p_t], dtype=torch.int64, device=dev)
    if use_dist:
        dist.all_reduce(pairs)
    total_pairs = int(pairs.item())
    ms_per_step = elapsed / timed_steps * 1e3
    value = total_pairs * timed_steps / elapsed / 1e6

    workload = (("%s: %d videos x %d frames x %d dets/frame, %d categories " +
                 ("split by video over the GPUs" if strong else "per GPU") +
                 "%s; LVISEval + TaoEval passes")
                % (CONFIGS[args.config]["name"] if (args.videos, args.frames, args.dets)
                   == tuple(CONFIGS[args.config][k] for k in ("videos", "frames", "dets"))
                   else "SYNTH custom", args.videos, args.frames, args.dets,
                   args.cats, ", decimal coordinates" if args.decimal else ""))

    # ---- per-kernel durations inside the timed steps -> dominant kernel
    stages, roof, roof_other, kernels_ms, step_roof = None, None, None, None, None
    if rank == 0:
        stages = engine.time_stages(dpl, wsl, dpt, wst, reps=10)
        models, strict = {}, {}
        for side, dp, ws in (("lvis", dpl, wsl), ("tao", dpt, wst)):
            mm, gg, vv = kernel_models(dp, ws)
            for k, v in mm.items():
                models[side + ":" + k] = (v, gg.get(k), vv.get(k))
            for k, v in survey_models(dp, ws).items():
                strict[side + ":" + k] = v
        cands, kernels_ms = [], {}
        probed = max(1, len(range(PROBE_EVERY // 2, timed_steps, PROBE_EVERY)))
        for name, (tot, calls) in in_step.items():
            k_ms = tot / calls
            kernels_ms[name] = round(k_ms, 4)
            alg, grid, variant = models.get(name, (None, None, None))
            ent = {"bound": "hbm", "kernel": name, "achieved": None,
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                   "traffic": None, "alg_bytes_per_launch": alg,
                   "kernel_ms": round(k_ms, 4),
                   "launches_per_step": round(calls / probed, 2),
                   "timed": "HIP events on the kernel's stream inside the timed "
                            "steps (every %dth step, %d launches)"
                            % (PROBE_EVERY, calls)}
            if name in strict:
                # SURVEY 8(d)'s bytes alone (no dst[] / dt_meta / plans): the
                # figure comparable round to round and with the survey
                sv = strict[name]
                ent["alg_bytes_survey"] = int(sv)
                ent["frac_survey"] = round(sv / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            if alg:
                ach = alg / (k_ms * 1e-3) / 1e9
                ent["achieved"] = round(ach, 2)
                ent["frac"] = round(ach / HBM_PEAK_GBS, 5)
                tr = pmc_traffic(name.split(":", 1)[1], variant, grid, workload)
                if tr and tr.get("bytes") is not None:
                    ent["traffic"] = tr["bytes"]
                    ent["traffic_source"] = tr["source"]
                elif tr:
                    ent["traffic_stale"] = tr["stale"]
            if name in alone:
                # the same kernel with nothing beside it (serial steps after
                # the timed region): in-step durations of kernels off the
                # critical chain are mostly queueing
                a_ms = alone[name][0] / alone[name][1]
                ent["alone"] = {"kernel_ms": round(a_ms, 4),
                                "achieved": round(alg / (a_ms * 1e-3) / 1e9, 2) if alg else None,
                                "frac": round(alg / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                                if alg else None,
                                "frac_survey": round(strict[name] / (a_ms * 1e-3) / 1e9
                                                     / HBM_PEAK_GBS, 5)
                                if name in strict else None}
            cands.append(ent)
        # The dominant kernel = the longest launch INSIDE THE TIMED STEPS (the
        # in-step events; what rocprofv3's kernel table of the same command
        # ranks first), `alone` beside it -- since round 5 (round
        # 4 ranked by the alone durations, which picked another kernel than the
        # rounds before).  One exclusion, stated in `dominant_by`: a launch whose
        # in-step duration is more than twice its duration alone is queueing
        # for wave slots behind another stream's kernel, not work (round 4:
        # ss_split_kernel, 0.06 ms of work inside 0.32 ms).
        def is_work(c):
            a = c.get("alone")
            return a is None or a["kernel_ms"] * 2.0 >= c["kernel_ms"]
        cands.sort(key=lambda c: -c["kernel_ms"])
        ranked = [c for c in cands if is_work(c)] or cands
        if cands:
            # (the two longest launches of the default workload -- the 3D IoU
            # and the image level's match -- lie within 1-2 % of each other and
            # change places from box to box: among launches within 3 % of the
            # longest the one with the LOWER fraction is named, so that the
            # line names the same kernel run after run and errs low)
            near = [c for c in ranked if c["kernel_ms"] >= 0.97 * ranked[0]["kernel_ms"]
                    and c.get("frac") is not None]
            roof = min(near, key=lambda c: c["frac"]) if near else ranked[0]
            roof["dominant_by"] = ("longest average launch inside the timed steps "
                                   "(launches that take more than 2x their time alone "
                                   "-- queueing, not work -- excluded: %s; among "
                                   "launches within 3 %% of the longest -- %s -- the "
                                   "lower fraction)"
                                   % (", ".join(c["kernel"] for c in cands[:6]
                                                if not is_work(c)),
                                      ", ".join("%s %.4f ms" % (c["kernel"], c["kernel_ms"])
                                                for c in (near or ranked[:1]))))
            roof_other = [c for c in ranked if c is not roof][:4]
        if not use_dist:
            b = step_algorithmic_bytes(dpl, dpt)
            ach = b / (ms_per_step * 1e-3) / 1e9
            step_roof = {"alg_bytes_per_step": int(b), "achieved": round(ach, 2),
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                         "note": "SURVEY 8(d) B_alg of both passes / ms_per_step"}

    # ---- multi-GPU: every rank checks the assembled tables against what the
    # owners computed.  Own block: the plain single-GPU pass over this rank's
    # shard must give the same precision / recall block, bit for bit.  Other
    # blocks: a checksum of checksums -- each owner publishes the wrapping
    # int64 sum of its block's bit patterns, every rank compares it with the
    # sum over that block of ITS assembled table.
    exchange_ok = None
    if by_category and not args.emulate:
        ok = True
        for ev, dp, ws in ((plan.lvis, dpl, wsl), (plan.tao, dpt, wst)):
            engine.run(dp, ws)
            k0, k1, Kb = ev.k0, ev.k1, ev.Kb
            own = torch.zeros(2, dtype=torch.int64, device=dev)
            if k1 > k0:
                ok &= bool(torch.equal(ws.precision[:, :, k0:k1], ev.precision[:, :, k0:k1]))
                ok &= bool(torch.equal(ws.recall[:, k0:k1], ev.recall[:, k0:k1]))
                own[0] = ws.precision[:, :, k0:k1].contiguous().view(torch.int64).sum()
                own[1] = ws.recall[:, k0:k1].contiguous().view(torch.int64).sum()
            sums = torch.zeros((world, 2), dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(sums, own)
            for r in range(world):
                a, b = min(r * Kb, dp.n_cat), min((r + 1) * Kb, dp.n_cat)
                if b > a:
                    ok &= int(ev.precision[:, :, a:b].contiguous().view(torch.int64).sum()) == int(sums[r, 0])
                    ok &= int(ev.recall[:, a:b].contiguous().view(torch.int64).sum()) == int(sums[r, 1])
            ev.check()
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        exchange_ok = bool(flag.item())
    elif use_dist and not args.emulate:
        # unit partition: all ranks must hold identical assembled tables
        ok = True
        for ev in (plan.lvis, plan.tao):
            own = torch.stack([ev.precision.view(torch.int64).sum(),
                               ev.recall.view(torch.int64).sum()])
            lo, hi = own.clone(), own.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            ok &= bool(torch.equal(lo, hi))
        exchange_ok = ok

    # ---- verification + CPU baseline (C oracle = "port"), rank 0, N=1
    cpu, cpu_all, verified, t_host_flatten = None, None, None, None
    if not use_dist and rank == 0 and not args.no_cpu:
        import orclib
        nv = min(args.cpu_sample_videos, args.videos)
        if args.frames * args.dets < 3000:        # stress shape: many tiny videos
            nv = min(args.videos, max(nv, 3000000 // max(args.frames * args.dets, 1)))
        if nv == args.videos:
            # the whole workload: the oracle works on host tables built by the
            # numpy statement of the flatten stage from the same columns
            sgt, sdt = gt, dt
        else:
            sgt, sdt = synth(seed=args.seed, V=nv, F=args.frames, C=args.cats,
                             dets_per_frame=args.dets, decimal=args.decimal)
        t0 = time.perf_counter()
        sfl = flatten.flatten_lvis(sgt, sdt)
        sdt.track_id, _ = flatten.make_track_ids_unique(sdt)
        sft = flatten.flatten_tao(sgt, sdt)
        t_host_flatten = time.perf_counter() - t0
        what = ("%d of the %d videos of the same workload (%d box pairs, %.1f s) "
                "through oracle/tao_oracle.c, %s")
        if args.leg is None:
            orclib.set_threads(1)
            t0 = time.perf_counter()
            ol = orclib.run_flat(sfl, detail=False)
            ot = orclib.run_flat(sft, detail=False)
            t_cpu = time.perf_counter() - t0
            sp = sfl.n_pairs + ot["pairs"]
            cpu = {"value": round(sp / t_cpu / 1e6, 4), "unit": "Mpair/s", "cores": 1,
                   "kind": "port",
                   "sample": what % (nv, args.videos, sp, t_cpu, "single thread")}
        cores = orclib.set_threads(0)
        t0 = time.perf_counter()
        ol2 = orclib.run_flat(sfl, detail=False)
        ot2 = orclib.run_flat(sft, detail=False)
        t_all = time.perf_counter() - t0
        orclib.set_threads(1)
        if args.leg is not None:
            # (a leg of another run: the oracle is the checker only, on all cores)
            ol, ot = ol2, ot2
        sp = sfl.n_pairs + ot["pairs"]
        same = (np.array_equal(ol2["precision"], ol["precision"])
                and np.array_equal(ot2["precision"], ot["precision"])
                and np.array_equal(ot2["iou"], ot["iou"]))
        cpu_all = {"value": round(sp / t_all / 1e6, 4), "unit": "Mpair/s",
                   "cores": cores, "kind": "port", "equals_single_thread": bool(same),
                   "sample": what % (nv, args.videos, sp, t_all,
                                     "OpenMP over cells / categories, %d threads" % cores)}
        if not args.no_verify and nv == args.videos:
            # the tensors left behind by the timed (overlapped) steps, and the
            # IoU matrix of the track level
            torch.cuda.synchronize()
            verified = bool(
                np.array_equal(wsl.precision.cpu().numpy(), ol["precision"])
                and np.array_equal(wsl.recall.cpu().numpy(), ol["recall"])
                and np.array_equal(wst.precision.cpu().numpy(), ot["precision"])
                and np.array_equal(wst.recall.cpu().numpy(), ot["recall"])
                and (engine.guarded_pairs(dpt, wst) > 0     # (set-order bits there)
                     or np.array_equal(wst.iou[:dpt.n_iou].cpu().numpy(), ot["iou"]))
                and np.array_equal(wsl.num_gt.cpu().numpy(), ol["num_gt"])
                and np.array_equal(wst.num_gt.cpu().numpy(), ot["num_gt"]))
        elif not args.no_verify:
            gl = engine.evaluate_flat(sfl, dev)
            gtt = engine.evaluate_flat(sft, dev)
            verified = bool(
                np.array_equal(gl["matched"], ol["matched"])
                and np.array_equal(gl["ignored"], ol["ignored"])
                and np.array_equal(gl["precision"], ol["precision"])
                and np.array_equal(gl["recall"], ol["recall"])
                and np.array_equal(gtt["iou"], ot["iou"])
                and np.array_equal(gtt["matched"], ot["matched"])
                and np.array_equal(gtt["precision"], ot["precision"])
                and np.array_equal(gtt["recall"], ot["recall"]))

    # ---- decimal coordinates: a sample of the same workload through the Python
    # oracle in the reference's own frame order (CPython sets), against the HIP
    # path with its guard -- the C oracle above adds frames in timeline order
    set_order_check = None
    if args.decimal and not use_dist and rank == 0 and not args.no_cpu \
            and not args.no_verify:
        from oracle import pyoracle
        nv = min(6, args.videos)
        sgt, sdt = synth(seed=args.seed, V=nv, F=args.frames, C=args.cats,
                         dets_per_frame=args.dets, decimal=True)
        t0 = time.perf_counter()
        gj, pj = sgt.to_json(), sdt.to_json()
        pyoracle.make_track_ids_unique(pj)
        po = pyoracle.tao_eval(gj, pj, frame_order="set")
        t_py = time.perf_counter() - t0
        sdt.track_id, _ = flatten.make_track_ids_unique(sdt)
        gg = engine.evaluate_flat(flatten.flatten_tao(sgt, sdt), dev)
        set_order_check = {
            "videos": nv, "python_oracle_s": round(t_py, 2),
            "precision_equal": bool(np.array_equal(
                gg["precision"].reshape(po["precision"].shape), po["precision"])),
            "recall_equal": bool(np.array_equal(
                gg["recall"].reshape(po["recall"].shape), po["recall"]))}

    # ---- the metric's second half: end-to-end wall-clock of the drop-in CLI
    # on this very workload as FILES (JSON -> parse -> tables -> kernels ->
    # summaries -> printed text).  Writing the files is not part of it.
    wall = None
    if not use_dist and rank == 0 and not args.no_wallclock:
        wall = wallclock_leg(gt, dt)

    # ---- the same workload with DECIMAL coordinates (what real prediction
    # files hold: the frame-order guard of the 3D IoU is active), 20 steps, in
    # a subprocess of this very script once this process's GPU work is done
    decimal_leg = None
    if (not use_dist and rank == 0 and not args.decimal and not args.no_decimal_leg
            and not args.no_cpu and not args.serial):
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--leg", "decimal",
               "--steps", "20", "--warmup", "3", "--config", args.config,
               "--videos", str(args.videos), "--frames", str(args.frames),
               "--dets", str(args.dets), "--cats", str(args.cats),
               "--seed", str(args.seed)]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            d = json.loads(line[-1]) if line else None
            if r.returncode != 0 or d is None:
                decimal_leg = {"error": "rc %d: %s" % (r.returncode, r.stderr[-300:])}
            else:
                g = d["frame_order_guard"]
                decimal_leg = {
                    "ms_per_step": d["ms_per_step"], "value": d["value"],
                    "timed_steps": d["timed_steps"], "guard_active": g["active"],
                    "guard_where": g["where"], "near_ulp": g["near_ulp"],
                    "guard_ms": g["ms_per_step"],
                    "near_pairs": g["near_threshold_pairs"],
                    "bit_exact": d["bit_exact_vs_oracle"],
                    "set_order_sample": g["set_order_sample"],
                    "look_back_timeouts": d.get("look_back_timeouts"),
                    "look_back_giveups": d.get("look_back_giveups"),
                    "workload": d["config"]["workload"],
                    "leg_wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:       # (the leg must not lose the bench line)
            decimal_leg = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        out = {
            "metric": "box-pair IoU+match throughput", "value": round(value, 3),
            "unit": "Mpair/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": args.scaling if use_dist else "weak",
            "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "pairs_per_step": total_pairs,
                       "lvis_pairs_rank0": p_l, "tao_pairs_rank0": p_t,
                       "boxes_in_rank0": n_boxes_in,
                       "detections_rank0": dpl.n_dt, "tracks_rank0": dpt.n_dt,
                       "cells_rank0": [dpl.n_cells, dpt.n_cells],
                       "parallelism": ("single GPU" if not use_dist else
                                       "%s-sharded x%d" % (args.shard, world))},
            "timed_steps": timed_steps,
            "timed_region_s": round(elapsed, 4),
            "roofline": roof, "roofline_other": roof_other,
            "step_roofline": step_roof,
            "hbm_delivers": hbm_delivers(),
            "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all,
            "kernels_ms": kernels_ms,
            "kernels_alone_ms": {k: round(t / c, 4) for k, (t, c) in alone.items()},
            "stages_ms": stages,
            "streams": "serial" if args.serial else "4 (image-level || track-level, range masks + num_gt all-reduce aside) + RCCL" if (use_dist and not by_category)
            else "4 (image-level || track-level, ranges/sort || IoU) + RCCL all_gather" if use_dist
            else "4 (image-level || track-level, ranges/sort || IoU)",
            "host_launch_ms_per_step": round(host_ms, 4),
            "wall_clock_s": wall,
            "bit_exact_vs_oracle": verified,
            "decimal": decimal_leg,
            "look_back_timeouts": look_back_timeouts,
            "look_back_giveups": look_back_giveups,
            "frame_order_guard": {
                "exact_terms": bool(dpt.exact_terms),
                "active": bool(dpt.guard_active()),
                "where": ("device (taoamd_track_iou_near + taoamd_track_iou_setorder, "
                          "no host round trip)" if dpt.guard_on_device else
                          "host" if dpt.guard_active() else None),
                "near_ulp": dpt.near_ulp,
                "near_threshold_pairs": engine.guarded_pairs(dpt, wst),
                "ms_per_step": (round(sum(kernels_ms.get("tao:" + k, 0.0) for k in
                                          ("track_iou_near_kernel",
                                           "track_iou_setorder_kernel")), 4)
                                if kernels_ms else None),
                "set_order_sample": set_order_check},
            "ranks_verified": ranks_verified,
            "exchange_verified": exchange_ok,
            "exchange_chunk_bytes": ([plan.lvis.chunk_bytes, plan.tao.chunk_bytes]
                                     if by_category else None),
            "host_s": {"generate": round(t_gen, 2), "flatten": round(t_flat, 3),
                       "flatten_on": "host (numpy)" if by_category else "device",
                       "upload": round(t_h2d, 3),
                       "numpy_flatten_for_the_oracle": (
                           round(t_host_flatten, 2) if cpu else None)},
        }
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
