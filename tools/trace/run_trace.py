"""Runs the image-level pass of Config 2 on the instrumented library
(tools/trace/build_trace_lib.py) and prints, per kernel, how long the phases of a
wavefront / workgroup take (us; wall_clock64 at 100 MHz)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TAOAMD_LIBRARY"] = os.path.join(ROOT, "tools", "trace", "_lib_trace.so")
import torch  # noqa: E402
from tao_amodal_amd import _lib, engine, flatten as fl  # noqa: E402
from tao_amodal_amd.synth import synth  # noqa: E402


def table(t, names, end_col):
    t0 = t[:, 0].min()
    print("   records %d, kernel span %.1f us, last start %.1f us"
          % (len(t), (t[:, end_col].max() - t0) / 100.0, (t[:, 0].max() - t0) / 100.0))
    for i, nm in enumerate(names):
        a, b = t[:, i], t[:, i + 1]
        ok = (a > 0) & (b > 0)
        v = (b[ok] - a[ok]) / 100.0
        if len(v):
            print("   %-28s mean %6.2f  p50 %6.2f  p90 %6.2f  p99 %6.2f  max %6.2f"
                  % (nm, v.mean(), *np.percentile(v, [50, 90, 99]), v.max()))
    tot = (t[:, end_col] - t[:, 0]) / 100.0
    print("   %-28s mean %6.2f  p50 %6.2f  p90 %6.2f  p99 %6.2f  max %6.2f"
          % ("whole", tot.mean(), *np.percentile(tot, [50, 90, 99]), tot.max()))


def dump(lib, name, n):
    buf = np.zeros(8 * n, np.uint64)
    fn = getattr(lib, name)
    fn.argtypes = [C.c_void_p]
    assert fn(buf.ctypes.data) == 0
    t = buf.reshape(-1, 8)
    return t[t[:, 7] > 0].astype(np.int64)


def main():
    gt, dt = synth(V=200, F=300, C=1203, dets_per_frame=50)
    f = fl.flatten_lvis(gt, dt)
    dp = engine.DeviceProblem(f, "cuda:0")
    ws = engine.Workspace(dp)
    for _ in range(3):
        engine.run(dp, ws)
    torch.cuda.synchronize()
    lib = _lib.load()
    print("Config 2, image level, kernels alone (engine.run), one record per wavefront")
    print("seg_tile_kernel (one record per workgroup = tile)")
    t = dump(lib, "taoamd_trace_tile", 16384)
    full = t[:, 6] >= 2800
    for sel, nm in ((full, "full tiles (2816 elements)"), (~full, "partial tiles")):
        print("  ", nm)
        table(t[sel], ["load scores", "radix passes (high word)", "inversion scan",
                       "repair passes"], 5)
    print("match_group_kernel")
    table(dump(lib, "taoamd_trace_match", 65536),
          ["run descriptor", "detection / GT rows", "masks, LDS, IoU", "candidates, cells",
           "prefix maximum", "output words", "sequential greedy"], 7)
    print("acc_emit_kernel")
    t = dump(lib, "taoamd_trace_emit", 16384)
    table(t, ["chunk -> category", "set-up loads, LDS", "threshold search", "zero tail",
              "walk + emission"], 5)
    walk = (t[:, 5] - t[:, 4]) / 100.0
    for j in range(4):
        m = t[:, 6] == j
        print("   chunk %d of its category: %5d wavefronts, walk mean %5.1f max %5.1f"
              % (j, m.sum(), walk[m].mean(), walk[m].max()))
    m = t[:, 6] >= 4
    print("   later chunks           : %5d wavefronts, walk mean %5.1f max %5.1f"
          % (m.sum(), walk[m].mean(), walk[m].max()))


if __name__ == "__main__":
    main()
