#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X evaluation hot path.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input that
is already resident in HBM: for BOTH evaluators (image-level LVISEval and
track-level TaoEval) range masks -> (category, -score) sort -> [3D track IoU]
-> IoU + greedy match at 10 thresholds x {6 | 20} ranges -> accumulate
(precision[T,R,K,A] and recall materialised in the reference layout).

Workload at N=1: BASELINE.json configs[1], "Synthetic 200 videos x 300 frames
x 50 dets" with 1203 categories (SURVEY.md 8(d) Config 2).  For N>1 every
rank evaluates its own 200-video shard (weak scaling): match runs per rank,
then one RCCL exchange routes each category's records to its owner rank,
which sorts and accumulates them; an all-reduce(max) assembles the tensors.

Prints ONE JSON line on rank 0 (see the driver contract in the task text).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--videos", type=int, default=200)
    p.add_argument("--frames", type=int, default=300)
    p.add_argument("--dets", type=int, default=50)
    p.add_argument("--cats", type=int, default=1203)
    p.add_argument("--seed", type=int, default=20240807)
    p.add_argument("--cpu-sample-videos", type=int, default=200,
                   help="videos of the same workload timed through the C "
                        "oracle for cpu_baseline (rank 0, N=1 only)")
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--graph", action="store_true",
                   help="replay the step from captured hipGraphs instead of "
                        "launching the kernels (experimental: slower than the "
                        "stream launches on one GPU, see DESIGN.md)")
    p.add_argument("--serial", action="store_true",
                   help="run the two evaluator passes back to back on one stream")
    p.add_argument("--shard", choices=["category", "unit"], default="category",
                   help="multi-GPU partition: contiguous category blocks (no "
                        "record exchange) or the ranks' own videos (records "
                        "meet at the category owners)")
    p.add_argument("--emulate", default=None, metavar="W:R",
                   help="with --force-dist on one GPU: run the per-rank work of "
                        "rank R of a W-rank category-sharded job (diagnostic; "
                        "the printed value counts this rank's pairs only)")
    p.add_argument("--force-dist", action="store_true",
                   help="take the multi-GPU code path even with one rank")
    return p.parse_args()


def algorithmic_bytes_match(dp):
    """Compulsory HBM traffic of ONE launch of the fused LVIS IoU+match
    kernel (DESIGN.md 'Kernels'): boxes 32 B, range masks 4 B, flags 1 B per
    detection and GT; scatter index 4 B, cell index 4 B and 2 x 8 B output
    words per detection; 2 x 4 B CSR entries per cell."""
    return (dp.n_dt * (32 + 4 + 1 + 4 + 4 + 16 * dp.n_words)
            + dp.n_gt * (32 + 4 + 1) + dp.n_cells * 8)


def algorithmic_bytes_track_iou(dp):
    """ONE launch of the 3D-IoU kernel: every frame of every track once
    (4 B timeline position + 32 B box), 4 B per track offset, 8 B per pair
    written, 20 B of cell tables per cell."""
    frames = dp.t["dt_frame_pos"].numel() + dp.t["gt_frame_pos"].numel()
    return frames * 36 + (dp.n_dt + dp.n_gt) * 4 + dp.n_iou * 8 + dp.n_cells * 20


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the newest committed
    PMC summary (profiles/*_pmc.json, written by tools/prof_summary.py from
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very
    command); None when no profile is present."""
    import glob
    import re

    def version(path):      # r01_v11_pmc.json -> (1, 11): numeric, not lexical
        m = re.search(r"r(\d+)_v(\d+)_pmc", os.path.basename(path))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), key=version)
    if not files:
        return None
    with open(files[-1]) as f:
        ks = json.load(f)["kernels"]
    for name, v in ks.items():
        if kernel_substr in name:
            return v["hbm_bytes_corrected"]
    return None


def main():
    args = parse()
    # stdout carries exactly one JSON line: RCCL and the HIP runtime print
    # banners to the C-level stdout, so fd 1 is pointed at stderr for the run
    # and the line goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.force_dist:
        # the category-partitioned step keeps 4 compute streams + the RCCL
        # stream busy; with the default of 4 hardware queues per process they
        # alias and serialise (measured 1.25 -> 0.94 ms/step with 8)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from tao_amodal_amd import engine, flatten
    from tao_amodal_amd.synth import synth

    by_category = use_dist and args.shard == "category"
    data_world, data_rank = world, rank
    if args.emulate:
        assert by_category and world == 1, "--emulate needs --force-dist on one GPU"
        data_world, data_rank = (int(x) for x in args.emulate.split(":"))
    t0 = time.time()
    if by_category:
        # weak scaling: the data set grows with the number of ranks (one
        # 200-video shard per rank) and every rank evaluates its category
        # block of the WHOLE set, so the work per GPU stays fixed
        from tao_amodal_amd.columns import DTColumns, GTColumns
        parts = [synth(seed=args.seed + r, V=args.videos, F=args.frames,
                       C=args.cats, dets_per_frame=args.dets,
                       video_id_base=r * args.videos)
                 for r in range(data_world)]
        gt = GTColumns.concat([p[0] for p in parts])
        dt = DTColumns.concat([p[1] for p in parts])
        del parts
    else:
        gt, dt = synth(seed=args.seed + rank, V=args.videos, F=args.frames,
                       C=args.cats, dets_per_frame=args.dets,
                       video_id_base=rank * args.videos)
    t_gen = time.time() - t0
    t0 = time.time()
    fl = flatten.flatten_lvis(gt, dt)
    dt.track_id, _ = flatten.make_track_ids_unique(dt)
    ft = flatten.flatten_tao(gt, dt)
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
        if args.graph:
            # the launch sequence of a step is fixed: capture it once (stream
            # forks become parallel branches), a step is two hipGraph launches
            graphed = engine.GraphedPair(overlap, dpl, wsl, dpt, wst)
            step = graphed.run

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    probe = None
    if not use_dist and not args.serial and not args.graph:
        probe = engine.StageProbe()      # events on the kernels' streams
    t0 = time.perf_counter()
    for i in range(args.steps):
        # (the two dominant kernels are bracketed with events on every 4th
        # step only: an event pair costs a few microseconds of stream time)
        engine.PROBE = probe if i % 4 == 0 else None
        step()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3   # launch side only
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    engine.PROBE = None
    in_step_ms = probe.mean_ms() if probe is not None else {}
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- pairs: exact counts from the cell tables / the kernel's counter
    p_l = dpl.n_pairs
    if args.graph and not use_dist:   # the counter's memset node is unreliable under replay
        engine.stage_track_iou(dpt, wst)
        p_t = int(wst.pair_frames.item())
    else:
        p_t = int(plan.pair_frames()) if use_dist else int(wst.pair_frames.item())
    pairs = torch.tensor([p_l + p_t], dtype=torch.int64, device=dev)
    if use_dist:
        dist.all_reduce(pairs)
    total_pairs = int(pairs.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = total_pairs * args.steps / elapsed / 1e6

    # ---- stage breakdown + dominant-kernel roofline (HIP events on the
    # stream the kernels run on), measured outside the timed region
    stages, roof, roof_other = None, None, None
    if rank == 0:
        stages = engine.time_stages(dpl, wsl, dpt, wst, reps=max(args.steps, 10))
        # the two single-kernel stages; `roofline` reports the one that takes
        # longer (the dominant kernel of the step), `roofline_other` the other
        cands = []
        for name, sym, probe_key, iso_ms, alg in (
                ("match_group_kernel<fused> (LVIS box IoU + greedy match)",
                 "match_group_kernel<true>", "lvis:match", stages["lvis"]["match"],
                 algorithmic_bytes_match(dpl)),
                ("track_iou_dense_kernel (TAO 3D track IoU)",
                 "track_iou_dense_kernel", "tao:track_iou",
                 stages["tao"]["track_iou"], algorithmic_bytes_track_iou(dpt))):
            # launch duration inside the timed (overlapped) steps when it was
            # probed there, else the isolated stage time
            k_ms = in_step_ms.get(probe_key, iso_ms)
            ach = alg / (k_ms * 1e-3) / 1e9
            cands.append({"bound": "hbm", "kernel": name,
                          "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                          "traffic": pmc_traffic(sym),
                          "alg_bytes_per_launch": int(alg),
                          "kernel_ms": round(k_ms, 4),
                          "kernel_ms_isolated": round(iso_ms, 4),
                          "timed": "HIP events on the kernel's stream inside the "
                                   "timed steps" if probe_key in in_step_ms else
                                   "HIP events, stages run back to back after the "
                                   "timed steps"})
        cands.sort(key=lambda c: -c["kernel_ms"])
        roof, roof_other = cands[0], cands[1]

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

    # ---- verification + CPU baseline (C oracle = "port"), rank 0, N=1
    cpu, verified = None, None
    if not use_dist and rank == 0 and not args.no_cpu:
        import orclib
        nv = min(args.cpu_sample_videos, args.videos)
        sgt, sdt = synth(seed=args.seed, V=nv, F=args.frames, C=args.cats,
                         dets_per_frame=args.dets)
        sfl = flatten.flatten_lvis(sgt, sdt)
        sdt.track_id, _ = flatten.make_track_ids_unique(sdt)
        sft = flatten.flatten_tao(sgt, sdt)
        t0 = time.perf_counter()
        ol = orclib.run_flat(sfl, detail=False)
        ot = orclib.run_flat(sft, detail=False)
        t_cpu = time.perf_counter() - t0
        sp = sfl.n_pairs + ot["pairs"]
        cpu = {"value": round(sp / t_cpu / 1e6, 4), "unit": "Mpair/s", "cores": 1,
               "kind": "port",
               "sample": "%d of the %d videos of the same workload (%d box pairs, "
                         "%.1f s) through oracle/tao_oracle.c, single thread"
                         % (nv, args.videos, sp, t_cpu)}
        if not args.no_verify:
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
            if nv == args.videos:
                # the tensors left behind by the timed (overlapped) steps
                verified = verified and bool(
                    np.array_equal(wsl.precision.cpu().numpy(), ol["precision"])
                    and np.array_equal(wsl.recall.cpu().numpy(), ol["recall"])
                    and np.array_equal(wst.precision.cpu().numpy(), ot["precision"])
                    and np.array_equal(wst.recall.cpu().numpy(), ot["recall"]))

    if rank == 0:
        out = {
            "metric": "box-pair IoU+match throughput", "value": round(value, 3),
            "unit": "Mpair/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "SYNTH Config 2: %d videos x %d frames x %d "
                                   "dets/frame, %d categories per GPU; LVISEval + "
                                   "TaoEval passes" % (args.videos, args.frames,
                                                       args.dets, args.cats),
                       "pairs_per_step": total_pairs,
                       "lvis_pairs_rank0": p_l, "tao_pairs_rank0": p_t,
                       "detections_rank0": dpl.n_dt, "tracks_rank0": dpt.n_dt,
                       "cells_rank0": [dpl.n_cells, dpt.n_cells],
                       "parallelism": ("single GPU" if not use_dist else
                                       "%s-sharded x%d" % (args.shard, world))},
            "roofline": roof, "roofline_other": roof_other, "cpu_baseline": cpu,
            "stages_ms": stages, "streams": "serial" if args.serial else "2 (image-level || track-level)" if (use_dist and not by_category)
            else "4 (image-level || track-level, ranges/sort || IoU) + RCCL all_gather" if use_dist
            else "4 (image-level || track-level, ranges/sort || IoU)",
            "host_launch_ms_per_step": round(host_ms, 4),
            "hip_graph": bool(args.graph and not args.serial and not use_dist),
            "bit_exact_vs_oracle": verified,
            "exchange_verified": exchange_ok,
            "exchange_chunk_bytes": ([plan.lvis.chunk_bytes, plan.tao.chunk_bytes]
                                     if by_category else None),
            "host_s": {"generate": round(t_gen, 2), "flatten": round(t_flat, 2),
                       "upload": round(t_h2d, 2)},
        }
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
