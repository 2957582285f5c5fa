"""ctypes binding of the C ABI declared in include/tao_amodal_hip.h.

The shared library is built in-tree by ``tao_amodal_amd/csrc/build.sh`` (or
``__graft_entry__.build()``).  There is NO fallback: if the library is missing
or fails to load, importing the product path raises -- results never come
from a CPU path.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# TAOAMD_LIBRARY: another build of the same C ABI (A/B timing of kernel
# variants on one GPU box); the default is the in-tree library
SO_PATH = os.environ.get("TAOAMD_LIBRARY") or os.path.join(HERE, "libtao_amodal_hip.so")

OK = 0
ERR_HIP, ERR_ARG, JSON_FALLBACK = 1, 2, 16
N_THR, N_REC = 10, 101
LVIS_RNG, TAO_RNG = 6, 20
MAX_GT_PER_CELL = 3072
SEGMENT_TILE = 2816

_vp, _i64, _i32, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_size_t

SIGNATURES = {
    "taoamd_strerror": (C.c_char_p, [C.c_int]),
    "taoamd_last_error": (C.c_char_p, []),
    "taoamd_version": (C.c_int, []),
    "taoamd_thresholds_host": (C.c_int, [_vp, _vp]),
    "taoamd_set_thresholds": (C.c_int, [_vp, _vp]),
    "taoamd_set_ranges": (C.c_int, [_vp, _vp, _vp]),
    "taoamd_kernel_timing_enable": (C.c_int, [C.c_int]),
    "taoamd_kernel_timing_label": (C.c_int, [C.c_char_p]),
    "taoamd_kernel_timing_collect": (C.c_int, [_vp, _sz, _vp, _vp, _i32, _vp]),
    "taoamd_bb_iou": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "taoamd_bb_iou_host": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "taoamd_lvis_ranges": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _i64, _vp,
                                     _i32, _vp, _vp, _vp, _vp]),
    "taoamd_tao_ranges": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                    _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "taoamd_sort_segments_workspace": (_sz, [_i64]),
    "taoamd_sort_segments": (C.c_int, [_i64, _i32, _vp, _vp, _i32, _i32, _vp,
                                       _vp, _vp, _vp, _vp, _sz, _vp]),
    "taoamd_sort_plan_host": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "taoamd_sort_sampled_workspace": (_sz, [_i64, _i64, _i32]),
    "taoamd_sort_sampled_cap_limit": (C.c_int, [_i32]),
    "taoamd_sort_sampled_notify": (C.c_int, [_vp]),
    "taoamd_event_create": (C.c_int, [_vp]),
    "taoamd_event_destroy": (C.c_int, [_vp]),
    "taoamd_stream_wait_event": (C.c_int, [_vp, _vp]),
    "taoamd_sort_sampled": (C.c_int, [_i64, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _vp,
                                      _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp,
                                      _sz, _vp]),
    "taoamd_track_iou": (C.c_int, [_i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "taoamd_track_iou_single": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32,
                                          _vp, _vp, _vp]),
    "taoamd_track_iou_planned": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _i32, _vp, _vp, _vp]),
    "taoamd_track_stream": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "taoamd_json_pred_workspace": (_sz, [_sz]),
    "taoamd_json_pred_open": (_vp, [C.c_char_p, _vp, _sz, _vp, C.c_char_p, _sz, _vp]),
    "taoamd_json_pred_count": (_i64, [_vp]),
    "taoamd_json_pred_read": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _i32, _vp]),
    "taoamd_json_pred_convert": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _i32, _vp]),
    "taoamd_json_pred_close": (None, [_vp]),
    "taoamd_track_pad": (C.c_int, [_i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64,
                                   _vp, _vp, _vp]),
    "taoamd_track_iou_near": (C.c_int, [_i64, _vp, _vp, _i64, _vp, _i32, _i32,
                                        _vp, _vp, _vp]),
    "taoamd_track_iou_setorder_table": (_i64, [_i64, _i64]),
    "taoamd_track_iou_setorder": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                            _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32,
                                            _vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "taoamd_set_order_iou_host": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _vp,
                                            _i32, _vp]),
    "taoamd_pyset_union_order_host": (C.c_int, [_i64, _vp, _i64, _vp, _vp, _vp]),
    "taoamd_track_iou_plan_host": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp,
                                             _vp, _vp, _vp, _vp]),
    "taoamd_flat_map": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "taoamd_flat_count_bad": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "taoamd_flat_ordscore": (C.c_int, [_i64, _vp, _vp, _vp, _i32, _vp, _vp]),
    "taoamd_flat_ordinal": (C.c_int, [_i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "taoamd_flat_merge_cat": (C.c_int, [_i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp,
                                        _vp, _vp]),
    "taoamd_flat_split64": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp]),
    "taoamd_flat_compose": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "taoamd_flat_gather_cols": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp]),
    "taoamd_flat_track_of": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "taoamd_flat_keys": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp]),
    "taoamd_flat_track_kept": (C.c_int, [_i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp,
                                         _vp, _vp, _vp]),
    "taoamd_flat_track_sel": (C.c_int, [_i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp,
                                        _vp, _vp, _vp, _vp, _vp]),
    "taoamd_flat_track_filter": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp,
                                           _vp, _vp, _vp, _vp, _vp, _vp]),
    "taoamd_flat_frames": (C.c_int, [_i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _vp, _vp]),
    "taoamd_flat_scan": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "taoamd_flat_runs_by": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                      _vp]),
    "taoamd_flat_runs64_by": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _sz, _vp]),
    "taoamd_flat_rank_drop": (C.c_int, [_i64, _vp, _vp, _vp, _i32, _vp, _vp]),
    "taoamd_flat_filter": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _i32, _i64,
                                     _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                                     _vp, _vp]),
    "taoamd_flat_gather": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp,
                                     _vp, _vp, _vp, _vp, _vp, _vp]),
    "taoamd_flat_runs_workspace": (_sz, [_i64]),
    "taoamd_flat_runs": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "taoamd_flat_remap": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "taoamd_match_plan_host": (C.c_int, [_i64, _vp, _vp, _i32, _i32, _i32, _vp,
                                         _vp, _vp]),
    "taoamd_match": (C.c_int, [_i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32,
                               _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp,
                               _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp]),
    "taoamd_gather_rows": (C.c_int, [_i64, _i32, _vp, _vp, _i64, _vp, _vp,
                                     _vp, _vp]),
    "taoamd_compact_elems": (_sz, [_i32, _i32]),
    "taoamd_accumulate_compact": (C.c_int, [_i64, _i32, _i32, _vp, _vp, _vp,
                                            _vp, _i32, _i32, _i32, _vp, _vp,
                                            _vp, _sz, _vp]),
    "taoamd_rle_iou_workspace": (_sz, [_i64, _i64, _i64, _i64]),
    "taoamd_rle_iou": (C.c_int, [_i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp,
                                 _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _sz, _vp]),
    "taoamd_finalize": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "taoamd_exchange_chunk_bytes": (_sz, [_i32, _i32, _i64]),
    "taoamd_exchange_workspace": (_sz, [_i32, _i32, _i32]),
    "taoamd_exchange_sizes": (C.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "taoamd_exchange_pack": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp,
                                       _vp, _vp, _i64, _vp, _vp, _sz, _i32, _vp]),
    "taoamd_exchange_unpack": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _i64, _vp,
                                         _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    "taoamd_exchange_scores": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "taoamd_exchange_positions": (C.c_int, [_i64, _i32, _i32, _vp, _vp, _i32, _vp, _vp,
                                            _vp, _vp, _vp]),
    "taoamd_exchange_place": (C.c_int, [_i64, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp,
                                        _vp]),
    "taoamd_sort_workspace": (_sz, [_i64]),
    "taoamd_sort_by_cat_score": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _sz,
                                           _vp]),
    "taoamd_accumulate_workspace": (_sz, [_i64, _i32, _i32]),
    "taoamd_accumulate": (C.c_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32,
                                    _vp, _vp, _vp, _sz, _vp]),
    "taoamd_accumulate_error": (C.c_int, [_vp, _vp, _vp]),
    "taoamd_accumulate_chunked": (C.c_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32,
                                            _vp, _vp, _vp, _sz, _vp]),
    "taoamd_accumulate_compact_chunked": (C.c_int, [_i64, _i32, _i32, _vp, _vp, _vp,
                                                    _vp, _i32, _i32, _i32, _vp, _vp,
                                                    _vp, _sz, _vp]),
    "taoamd_accumulate_sweep_mode": (C.c_int, [_i32]),
    "taoamd_accumulate_plan_kind": (C.c_int, [_i64, _i32, _i32]),
    "taoamd_accumulate_spin_limit": (C.c_int, [_i32]),
    "taoamd_accumulate_giveup_counter": (C.c_int, [_vp]),
    "taoamd_accumulate_prepare": (C.c_int, [_i64, _i32, _i32, _vp, _i32, _vp, _sz, _vp]),
    "taoamd_accumulate_prepared": (C.c_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32,
                                    _vp, _vp, _vp, _sz, _vp]),
    "taoamd_accumulate_by_order": (C.c_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp,
                                             _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None


class TaoAmdError(RuntimeError):
    pass


def load():
    """Load the library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64/libhsa-runtime64.  It must be the HIP
    # runtime of the process: loading ours first would map a second runtime
    # (the system one named in our DT_NEEDED) that cannot see torch's device
    # allocations ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    if not os.path.exists(SO_PATH):
        raise TaoAmdError(
            "HIP extension not built: %s is missing.  Run "
            "tao_amodal_amd/csrc/build.sh (or __graft_entry__.build()).  "
            "There is no CPU fallback." % SO_PATH)
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


TIMING = False     # kernel_timing(True): engine passes label their launches


def kernel_timing(on):
    """Switch the library's per-kernel event timing on / off."""
    global TIMING
    TIMING = bool(on)
    check(load().taoamd_kernel_timing_enable(int(TIMING)),
          "taoamd_kernel_timing_enable")


def kernel_timing_label(label):
    load().taoamd_kernel_timing_label(label.encode() if label else None)


def kernel_timings():
    """{kernel name: (total ms, launches)} recorded since the last call
    (synchronises with the recorded events)."""
    import numpy as np
    names = C.create_string_buffer(8192)
    ms = np.zeros(128)
    calls = np.zeros(128, dtype=np.int64)
    n = C.c_int32(0)
    check(load().taoamd_kernel_timing_collect(
        C.addressof(names), len(names), ms.ctypes.data, calls.ctypes.data, 128,
        C.addressof(n)), "taoamd_kernel_timing_collect")
    out, raw = {}, names.raw.split(b"\0")
    for k in range(n.value):
        out[raw[k].decode()] = (float(ms[k]), int(calls[k]))
    return out


def set_constants(iou_thrs=None, rec_thrs=None, visibility_rng=None, area_rng=None,
                  time_rng=None):
    """Replace the calling thread's evaluation constants (taoamd_set_thresholds,
    taoamd_set_ranges); None = the reference's default for that table."""
    import numpy as np

    def arr(a, shape):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(shape)
        return a, a.ctypes.data
    lib = load()
    i, ip = arr(iou_thrs, (N_THR,))
    r, rp = arr(rec_thrs, (N_REC,))
    check(lib.taoamd_set_thresholds(ip, rp), "taoamd_set_thresholds")
    v, vp = arr(visibility_rng, (5, 2))
    a, ap = arr(area_rng, (5, 2))
    t, tp = arr(time_rng, (4, 2))
    check(lib.taoamd_set_ranges(vp, ap, tp), "taoamd_set_ranges")


SWEEP_MODES = {"auto": -1, "chunked": 0, "lookback": 1, "twopass": 2}


def sweep_mode(mode="auto", spin_limit=0):
    """Which sweep long categories take, for the process
    (taoamd_accumulate_sweep_mode: "auto" | "chunked" | "lookback" | "twopass"),
    and the look-back's poll limit (taoamd_accumulate_spin_limit; 0 default,
    < 0 every look-back gives up at once)."""
    lib = load()
    check(lib.taoamd_accumulate_sweep_mode(SWEEP_MODES[mode]), "taoamd_accumulate_sweep_mode")
    check(lib.taoamd_accumulate_spin_limit(int(spin_limit)), "taoamd_accumulate_spin_limit")


def check(status, what):
    if status != OK:
        lib = load()
        raise TaoAmdError("%s failed: %s [%s]" % (
            what, lib.taoamd_strerror(status).decode(),
            lib.taoamd_last_error().decode()))
