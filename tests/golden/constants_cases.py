"""The edits of ``params`` that tests/golden/make_golden_constants.py applies to
the reference evaluators and tests/test_gpu_constants.py to this repository's
(no import of the reference here: the GPU box does not have it)."""
import numpy as np


def cases():
    """{case: {attribute of params: value}}; "lvis:" / "tao:" prefixes mark the
    attributes only one of the two Params classes has."""
    return {
        # fewer thresholds than the kernels' blocks, not ascending
        "few": {"iou_thrs": np.array([0.75, 0.3, 0.9, 0.5]),
                "rec_thrs": np.linspace(0.0, 1.0, 11)},
        # more than one block of either
        "many": {"iou_thrs": np.linspace(0.05, 0.95, 19),
                 "rec_thrs": np.linspace(0.0, 1.0, 201)},
        # recall thresholds out of order: the reference stops filling a row at
        # the first threshold the category never reaches
        "unsorted_rec": {"rec_thrs": np.array([0.9, 0.1, 0.5, 1.0, 0.0, 0.3, 0.7])},
        # other bounds, same number of ranges
        "ranges": {"lvis:visibility_rng": [[0, 1.0], [0, 0.3], [0.3, 0.6], [0.6, 1.0],
                                           [0.05, 0.9], [0, 1.0]],
                   "tao:area_rng": [[0, 1e10], [0, 2000.0], [2000.0, 20000.0],
                                    [20000.0, 1e10], [500.0, 1e10]],
                   "tao:time_rng": [[0, 1e5], [0, 5], [5, 12], [12, 1e5]]},
        # another NUMBER of ranges: the reference loops over whatever the lists
        # hold, the last visibility range is the out-of-frame one, the last
        # area range the occlusion one (L/eval.py:140-145, T/eval.py:271-276).
        # Fewer than the labels name: summarize() raises IndexError there
        "ranges3": {"lvis:visibility_rng": [[0, 1.0], [0.2, 0.7], [0, 1.0]],
                    "tao:area_rng": [[0, 1e10], [1500.0, 30000.0], [0, 1e10]],
                    "tao:time_rng": [[0, 1e5], [4, 11]]},
        # ... and more than one block of the kernels' range slots
        "ranges8": {"lvis:visibility_rng": [[0, 1.0], [0, 0.1], [0.1, 0.8], [0.8, 1.0],
                                            [0, 0.8], [0.05, 0.3], [0.3, 0.9], [0, 1.0]],
                    "tao:area_rng": [[0, 1e10], [0, 1024.0], [1024.0, 9216.0],
                                     [9216.0, 1e10], [0, 5000.0], [5000.0, 1e10],
                                     [100.0, 1e10]],
                    "tao:time_rng": [[0, 1e5], [0, 3], [3, 10], [10, 1e5], [2, 6],
                                     [6, 1e5]]},
    }


def edit(params, case, kind):
    for name, value in case.items():
        if ":" in name:
            side, name = name.split(":")
            if side != kind:
                continue
        setattr(params, name, value if isinstance(value, np.ndarray)
                else [list(v) for v in value])
