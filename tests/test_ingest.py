"""Native columnar prediction reader (csrc/ingest.cpp) against json.load."""
import json
import os

import numpy as np
import pytest

from goldenio import FIXTURES, path
from tao_amodal_amd.columns import DTColumns

HAVE = os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                   "tao_amodal_amd", "libtao_amodal_ingest.so"))
pytestmark = pytest.mark.skipif(not HAVE, reason="ingest library not built")


def _same(a, b):
    for f in DTColumns.FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_files_parse_identically(name):
    p = path(name, "pred.json")
    _same(DTColumns.from_file_native(p), DTColumns.from_json(json.load(open(p))))


def test_awkward_but_valid_json(tmp_path):
    rng = np.random.default_rng(0)
    preds = []
    for k in range(500):
        preds.append({
            "note": 'braces } ] { [ and "quotes\\" inside',
            "score": float(rng.random()) * 10 ** float(rng.integers(-12, 3)),
            "extra": {"nested": [1, 2, {"x": "}"}], "t": True, "n": None},
            "bbox": [float(rng.normal()) * 300, int(rng.integers(0, 700)),
                     float(rng.random()) * 1e-7, 12.5],
            "category_id": int(rng.integers(1, 1203)),
            "image_id": int(rng.integers(0, 2 ** 40)),
            "track_id": int(rng.integers(-3, 10 ** 12)),
            "video_id": int(rng.integers(0, 3000)),
            "segmentation": [[1.5, 2.5, 3.5]],
        })
    preds[3].pop("track_id")
    preds[3].pop("video_id")
    preds[7]["image_id"] = 2 ** 62 + 12345          # beyond double precision
    p = tmp_path / "p.json"
    p.write_text(json.dumps(preds, indent=2))
    a = DTColumns.from_file_native(str(p))
    _same(a, DTColumns.from_json(preds))
    assert a.image_id[7] == 2 ** 62 + 12345 and a.track_id[3] == -1
    p.write_text(json.dumps(preds, separators=(",", ":")))
    _same(DTColumns.from_file_native(str(p)), DTColumns.from_json(preds))
    p.write_text("[]")
    assert len(DTColumns.from_file_native(str(p))) == 0


def test_errors(tmp_path):
    p = tmp_path / "p.json"
    p.write_text('{"a": 1}')
    with pytest.raises(AssertionError):
        DTColumns.from_file_native(str(p))
    p.write_text('[{"image_id": 1, "category_id": 2, "bbox": [1, 2, 3], "score": 1}]')
    with pytest.raises(ValueError):
        DTColumns.from_file_native(str(p))
    p.write_text('[{"image_id": 1, "category_id": 2, "score": 1}]')
    with pytest.raises(ValueError, match="bbox"):
        DTColumns.from_file_native(str(p))
    with pytest.raises(FileNotFoundError):
        DTColumns.from_file_native(str(tmp_path / "missing.json"))
