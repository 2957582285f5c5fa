"""cProfile of the drop-in CLI on the bench workload (development aid): where
the host spends the wall clock between the parse and the printed tables.
    python tools/cli_profile.py [--config 3s] [--serial]
Two warm-up runs first (torch, HIP context, allocator), then one profiled run
per mode: the levels one after the other on the main thread (every function
visible to the profiler), and the CLI as shipped (levels side by side) with
its own TAOAMD_TIMING split."""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import bench  # noqa: E402
from tao_amodal_amd.synth import synth  # noqa: E402


def main():
    a = bench.parse()
    d = tempfile.mkdtemp(prefix="taoamd_prof_")
    gt, dt = synth(seed=a.seed, V=a.videos, F=a.frames, C=a.cats, dets_per_frame=a.dets)
    gt.write_json(os.path.join(d, "gt.json"))
    dt.write_json(os.path.join(d, "pred.json"))
    del gt, dt
    import eval_on_tao_amodal as cli
    argv = ["--track_result", os.path.join(d, "pred.json"), "--annotation",
            os.path.join(d, "gt.json"), "--output_log", os.path.join(d, "eval.log")]
    os.environ["TAOAMD_TIMING"] = "1"
    sink = io.StringIO()

    def run():
        out = sys.stdout
        sys.stdout = sink
        try:
            t = time.perf_counter()
            cli.main(argv)
            return time.perf_counter() - t
        finally:
            sys.stdout = out
    for _ in range(2):
        print("warm-up run: %.3f s" % run(), file=sys.stderr)
    from tao_amodal_amd.evaluation import _core
    for serial in (True, False):
        if serial:
            os.environ["TAOAMD_CLI_SERIAL"] = "1"
        else:
            os.environ.pop("TAOAMD_CLI_SERIAL", None)
        _core.TIMING.clear()
        pr = cProfile.Profile()
        pr.enable()
        t = run()
        pr.disable()
        print("\n==== %s: %.3f s" % ("levels one after the other" if serial else
                                       "as shipped (main thread only profiled)", t))
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
        print(s.getvalue()[:14000])
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
        print(s.getvalue()[:8000])


if __name__ == "__main__":
    main()
