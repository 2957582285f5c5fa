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
    with pytest.raises(KeyError, match="bbox"):      # as the reference's r["bbox"]
        DTColumns.from_file_native(str(p))
    with pytest.raises(FileNotFoundError):
        DTColumns.from_file_native(str(tmp_path / "missing.json"))


# ---------------------------------------------------------------- annotation file
from goldenio import load_inputs            # noqa: E402
from tao_amodal_amd.columns import GTColumns  # noqa: E402
from tao_amodal_amd.synth import synth      # noqa: E402


def _same_gt(a, b):
    for f in GTColumns.FIELDS:
        x, y = getattr(a, f), getattr(b, f)
        assert x.dtype == y.dtype and x.shape == y.shape, f
        assert np.array_equal(x, y, equal_nan=x.dtype.kind == "f"), f


@pytest.mark.parametrize("name", FIXTURES)
def test_annotation_fixtures_parse_identically(name, tmp_path):
    gt, _ = load_inputs(name)
    p = tmp_path / "gt.json"
    p.write_text(json.dumps(gt))
    _same_gt(GTColumns.from_file_native(str(p)), GTColumns.from_json(gt))
    p.write_text(json.dumps(gt, indent=1))
    _same_gt(GTColumns.from_file_native(str(p)), GTColumns.from_json(gt))


def test_annotation_awkward_but_valid(tmp_path):
    gt, _ = synth(seed=9, V=4, F=6, C=30, dets_per_frame=4, n_merged=3)
    d = gt.to_json()
    d["info"] = {"note": 'braces } ] { [ "quoted\\" text', "nested": [[{"a": "]"}]]}
    d["licenses"] = [{"id": 1, "url": "http://x/{y}"}]
    for k, a in enumerate(d["annotations"]):
        a["segmentation"] = [[1.5, 2, 3]]
        a["name"] = "ann } %d" % k
        # every truthiness flavour Python accepts for `if a.get("ignore", 0)`
        a["ignore"] = [0, 1, True, False, None, 0.0, 2.5, "", "x", [], [0], {}, {"a": 0}][k % 13]
        a["out_of_frame"] = [False, True, 0, 1, None, "yes", ""][k % 7]
        if k % 5 == 0:
            a["area"] = float(a["area"]) + 0.125
    for k, t in enumerate(d["tracks"]):
        t["ignore"] = [0, 1, True, None, "x"][k % 5]
        if k % 3 == 0:
            del t["ignore"]
    d["categories"][0].pop("frequency", None)
    d["images"][0]["frame_index"] = 3.5
    d["images"][1]["id"] = d["images"][1]["id"]           # untouched
    d["annotations"][2]["id"] = 2 ** 60 + 7                # beyond double precision
    p = tmp_path / "gt.json"
    p.write_text(json.dumps(d, indent=2))
    a = GTColumns.from_file_native(str(p))
    _same_gt(a, GTColumns.from_json(d))
    assert a.ann_id[2] == 2 ** 60 + 7 and a.cat_freq[0] == ord("?")
    assert a.cat_merged.shape[0] > 0


def test_annotation_errors(tmp_path):
    gt, _ = synth(seed=9, V=2, F=3, C=20, dets_per_frame=2)
    base = gt.to_json()
    p = tmp_path / "gt.json"
    p.write_text("[1, 2]")
    with pytest.raises(AssertionError, match="not supported"):
        GTColumns.from_file_native(str(p))
    with pytest.raises(FileNotFoundError):
        GTColumns.from_file_native(str(tmp_path / "missing.json"))
    d = dict(base)
    del d["tracks"]
    p.write_text(json.dumps(d))
    with pytest.raises(KeyError, match="tracks"):
        GTColumns.from_file_native(str(p))
    d = json.loads(json.dumps(base))
    del d["images"][1]["neg_category_ids"]
    p.write_text(json.dumps(d))
    with pytest.raises(KeyError, match="neg_category_ids"):
        GTColumns.from_file_native(str(p))
    d = json.loads(json.dumps(base))
    del d["annotations"][0]["visibility"]
    p.write_text(json.dumps(d))
    with pytest.raises(KeyError, match="visibility"):
        GTColumns.from_file_native(str(p))
    p.write_text(json.dumps(base)[:-20])
    with pytest.raises(ValueError):
        GTColumns.from_file_native(str(p))


def test_annotation_classes_share_the_native_columns(tmp_path):
    """LVIS(path) / Tao(path) take the native reader and only parse the dict
    form of the file when ``dataset`` is asked for."""
    from tao_amodal_amd.evaluation.lvis_amodal import LVIS
    from tao_amodal_amd.evaluation.tao_amodal import Tao
    gt, _ = synth(seed=3, V=2, F=4, C=20, dets_per_frame=3)
    p = tmp_path / "gt.json"
    p.write_text(json.dumps(gt.to_json()))
    for cls in (LVIS, Tao):
        obj = cls(str(p))
        assert obj._dataset is None
        _same_gt(obj.columns, gt)
        assert obj.get_cat_ids() == gt.cat_id.tolist()
        assert obj._dataset is None
        assert len(obj.dataset["annotations"]) == len(gt.ann_id)


# ---------------------------------------------------------------- host sort
def test_native_sort_equals_numpy_lexsort():
    from tao_amodal_amd import flatten
    assert flatten._host_lib(), "host library not built"
    rng = np.random.default_rng(5)
    for n, nkeys, quant in ((60000, 7, 0), (200000, 50000, 20), (70000, 1, 3)):
        key = rng.integers(0, nkeys, n).astype(np.int64) * 1000003 - 17
        score = rng.random(n)
        if quant:
            score = np.round(score * quant) / quant
        score[rng.integers(0, n, 50)] = np.nan
        score[rng.integers(0, n, 50)] = -0.0
        score[rng.integers(0, n, 50)] = 0.0
        score[rng.integers(0, n, 20)] = np.inf
        want = np.lexsort((np.arange(n), -score, key))
        assert np.array_equal(flatten.sort_key_score(key, score), want)
        assert np.array_equal(flatten.sort_key_score(key),
                              np.argsort(key, kind="stable"))


def test_flatten_is_the_same_with_and_without_the_native_sort(monkeypatch):
    from tao_amodal_amd import flatten
    gt, dt = synth(seed=12, V=8, F=120, C=40, dets_per_frame=70, n_present=6)
    assert len(dt) > 50000
    a_l = flatten.flatten_lvis(gt, dt)
    ids, _ = flatten.make_track_ids_unique(dt)
    dt.track_id = ids
    a_t = flatten.flatten_tao(gt, dt)
    monkeypatch.setattr(flatten, "_HOST_LIB", False)
    dt._limit_cache = None
    b_l = flatten.flatten_lvis(gt, dt)
    b_t = flatten.flatten_tao(gt, dt)
    for a, b in ((a_l, b_l), (a_t, b_t)):
        assert a.keys() == b.keys()
        for k in a:
            if isinstance(a[k], (np.ndarray, flatten.LazyRows)):
                assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
            else:
                assert a[k] == b[k], k


def test_parallel_boundary_scan_with_hostile_strings(tmp_path):
    """Files above 1 MB are cut into per-thread chunks; strings full of
    quotes, backslash runs and brackets must not confuse the chunk logic."""
    gt, dt = synth(seed=2, V=6, F=100, C=40, dets_per_frame=60, n_present=6)
    preds = dt.to_json()
    for k in range(0, len(preds), 7):
        preds[k]["note"] = 'q"\\" } ] \\\\' + "\\" * ((k % 5) * 2) + ' [ {' + "x" * (k % 211)
    p = tmp_path / "p.json"
    p.write_text(json.dumps(preds))
    assert p.stat().st_size > (2 << 20)
    _same(DTColumns.from_file_native(str(p)), DTColumns.from_json(preds))
    d = gt.to_json()
    for k, a in enumerate(d["annotations"]):
        a["note"] = '\\\\"' * (k % 4) + "]}" + "y" * (k % 173)
    p = tmp_path / "g.json"
    p.write_text(json.dumps(d) + "   \n")
    _same_gt(GTColumns.from_file_native(str(p)), GTColumns.from_json(d))
    # trailing garbage after the list / truncated list
    p.write_text(json.dumps(preds)[:-1])
    with pytest.raises(ValueError):
        DTColumns.from_file_native(str(p))


def test_boundary_scan_with_chunks_starting_at_any_depth(tmp_path):
    """The bracket walk of a chunk does not know the depth it starts at
    (ingest.cpp ChunkWalk): elements with nested lists / objects several levels
    deep and hostile strings, cut into 64 KB chunks by 40 threads, must give
    the boundaries json.load gives.  A separate process: the thread count is
    read once per process."""
    import subprocess
    import sys
    rng = np.random.default_rng(5)
    _, dt = synth(seed=3, V=4, F=80, C=30, dets_per_frame=40, n_present=5)
    preds = dt.to_json()
    for k, q in enumerate(preds):
        depth = int(rng.integers(0, 5))
        v = [k, 'x]"{' + "\\" * (k % 3)]
        for lvl in range(depth):
            v = {"l%d" % lvl: [v, {"s": "}" * (k % 7)}]} if lvl % 2 else [v, [], {}]
        if k % 3:
            q["extra"] = v
        if k % 11 == 0:
            q["pad"] = "p" * int(rng.integers(0, 3000))
    p = tmp_path / "p.json"
    p.write_text(json.dumps(preds))
    assert p.stat().st_size > (2 << 20)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); "
            "from tao_amodal_amd.columns import DTColumns; "
            "d = DTColumns.from_file_native(%r); "
            "np.savez(%r, image_id=d.image_id, bbox=d.bbox, score=d.score, "
            "track_id=d.track_id)" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                      str(p), str(tmp_path / "out.npz")))
    for threads in ("40", "3"):
        subprocess.run([sys.executable, "-c", code], check=True,
                       env=dict(os.environ, TAOAMD_HOST_THREADS=threads))
        z = np.load(tmp_path / "out.npz")
        want = DTColumns.from_json(preds)
        assert np.array_equal(z["image_id"], want.image_id)
        assert np.array_equal(z["bbox"], want.bbox)
        assert np.array_equal(z["score"], want.score)
        assert np.array_equal(z["track_id"], want.track_id)


def test_boundary_scan_of_backslash_free_chunks(tmp_path):
    """Chunks without a backslash take the 64-bytes-at-a-time walk (quote /
    bracket bit masks, in-string mask as the running parity of the quotes):
    strings full of brackets, of every length around the block size, nested
    values, and chunk / block boundaries that fall inside strings."""
    import subprocess
    import sys
    rng = np.random.default_rng(6)
    _, dt = synth(seed=7, V=4, F=80, C=30, dets_per_frame=40, n_present=5)
    preds = dt.to_json()
    for k, q in enumerate(preds):
        q["note"] = "]}[{" * (k % 5) + "x" * int(rng.integers(0, 200)) + "{[" * (k % 3)
        if k % 4 == 0:
            q["deep"] = [[{"a": ["}]", {"b": "[" * (k % 70)}]}], "]" * (k % 130)]
    text = json.dumps(preds)
    assert "\\" not in text
    p = tmp_path / "p.json"
    p.write_text(text)
    assert p.stat().st_size > (2 << 20)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); "
            "from tao_amodal_amd.columns import DTColumns; "
            "d = DTColumns.from_file_native(%r); "
            "np.savez(%r, image_id=d.image_id, bbox=d.bbox, score=d.score, "
            "track_id=d.track_id)" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                      str(p), str(tmp_path / "out.npz")))
    want = DTColumns.from_json(preds)
    for threads in ("37", "2"):
        subprocess.run([sys.executable, "-c", code], check=True,
                       env=dict(os.environ, TAOAMD_HOST_THREADS=threads))
        z = np.load(tmp_path / "out.npz")
        assert np.array_equal(z["image_id"], want.image_id)
        assert np.array_equal(z["bbox"], want.bbox)
        assert np.array_equal(z["score"], want.score)
        assert np.array_equal(z["track_id"], want.track_id)


def test_reader_corner_cases_of_json_load(tmp_path):
    """What json.load accepts the reader accepts with the same values, what the
    evaluator cannot use it rejects: out-of-range exponents (inf / 0.0 as
    float() gives them), keys written with \\u escapes, a boolean score
    (True == 1), bare scalars between the objects, bytes after the list."""
    ok = tmp_path / "ok.json"
    ok.write_text('[{"\\u0069mage_id": 3, "category_id": 2, "bbox": [1e400, 1e-400, -1e999, 2],'
                  ' "score": true, "track_id": 1, "video_id": 1},\n'
                  ' {"image_id": 4, "category_id": 2, "bbox": [0, 0, 1, 1], "score": false}]  \n')
    got = DTColumns.from_file_native(str(ok))
    want = DTColumns.from_json(json.load(open(ok)))
    _same(got, want)
    assert np.isinf(got.bbox[0, 0]) and got.bbox[0, 1] == 0.0 and got.bbox[0, 2] == -np.inf
    assert got.score.tolist() == [1.0, 0.0] and got.image_id.tolist() == [3, 4]
    for text in ('[5, {"image_id": 4, "category_id": 2, "bbox": [0, 0, 1, 1], "score": 1}]',
                 '[{"image_id": 4, "category_id": 2, "bbox": [0, 0, 1, 1], "score": 1}, "x"]',
                 '[{"image_id": 4, "category_id": 2, "bbox": [0, 0, 1, 1], "score": 1}] tail'):
        bad = tmp_path / "bad.json"
        bad.write_text(text)
        with pytest.raises(ValueError):
            DTColumns.from_file_native(str(bad))


def test_native_writer_round_trips_the_columns(tmp_path):
    """csrc/jsonwrite.cpp: columns -> JSON text -> the same columns, through
    the native reader AND through json.load (decimal boxes, merged categories,
    shuffled ids, ignore flags)."""
    import json
    from tao_amodal_amd.synth import synth
    gt, dt = synth(seed=9, V=3, F=7, C=14, dets_per_frame=6, n_present=4,
                   decimal=True, shuffle_image_ids=True, n_merged=2)
    gt.trk_ignore[::5] = 1
    gt.ann_ignore[::7] = 1
    dt.score[:3] = [1.0, 0.0, 1e-7]
    gp, pp = str(tmp_path / "gt.json"), str(tmp_path / "pred.json")
    gt.write_json(gp)
    dt.write_json(pp)
    for g2 in (GTColumns.from_file_native(gp), GTColumns.from_json(json.load(open(gp)))):
        for f in GTColumns.FIELDS:
            assert np.array_equal(getattr(g2, f), getattr(gt, f)), f
    for d2 in (DTColumns.from_file_native(pp), DTColumns.from_json(json.load(open(pp)))):
        for f in DTColumns.FIELDS:
            assert np.array_equal(getattr(d2, f), getattr(dt, f)), f
    assert json.load(open(gp)) == gt.to_json()
    assert json.load(open(pp)) == dt.to_json()


def test_select_videos_is_the_subset_of_the_annotation_file():
    from tao_amodal_amd.synth import synth
    gt, _ = synth(seed=9, V=6, F=5, C=14, dets_per_frame=4, n_present=4)
    keep = np.array([0, 1, 0, 0, 1, 1], bool)
    sub = gt.select_videos(keep)
    j = gt.to_json()
    vids = set(gt.vid_id[keep].tolist())
    want = dict(j, videos=[v for v in j["videos"] if v["id"] in vids],
                images=[i for i in j["images"] if i["video_id"] in vids],
                tracks=[t for t in j["tracks"] if t["video_id"] in vids])
    imgs = {i["id"] for i in want["images"]}
    want["annotations"] = [a for a in j["annotations"] if a["image_id"] in imgs]
    assert sub.to_json() == want


def test_numbers_equal_pythons_float_bit_for_bit(tmp_path):
    """The reader's short-decimal fast path (<= 15 digits, no exponent: one
    exact division) and its general route give the double Python's float()
    gives for the same text -- random texts of 1..17 significant digits, with
    and without exponents, signs, leading / trailing zeros."""
    rng = np.random.default_rng(11)
    texts = ["0", "-0", "0.0", "-0.0", "0.000", "1", "-1", "10", "123456789012345",
             "1234567890123456", "0.1", "0.30000000000000004", "999999999999999",
             "99999999999999.9", "0.000000000000001", "0.0000000000000000000001",
             "4.35", "2.675", "1e3", "1E-3", "1.5e+10", "-2.5E-7", "1e22", "1e23",
             "123456789.123456789", "9007199254740993", "0.1e1", "5e-324", "1.7976931348623157e308"]
    for _ in range(4000):
        nd = int(rng.integers(1, 18))
        digits = "".join(str(int(d)) for d in rng.integers(0, 10, nd))
        cut = int(rng.integers(0, nd + 1))
        t = (digits[:cut] or "0") + ("." + digits[cut:] if cut < nd else "")
        if rng.random() < 0.3:
            t = "-" + t
        if rng.random() < 0.15:
            t += "e%d" % int(rng.integers(-30, 30))
        texts.append(t)
    rows = ['{"image_id": 1, "category_id": 1, "score": %s, "bbox": [%s, %s , %s\t,%s]}'
            % (texts[(k + 1) % len(texts)], texts[k], texts[(k + 2) % len(texts)],
               texts[(k + 3) % len(texts)], texts[(k + 4) % len(texts)])
            for k in range(len(texts))]
    p = tmp_path / "p.json"
    p.write_text("[" + ",\n".join(rows) + "]")
    d = DTColumns.from_file_native(str(p))
    want = np.array([float(t) for t in texts])
    n = len(texts)
    got = d.bbox[:, 0]
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert np.array_equal(d.score.view(np.uint64), np.roll(want, -1).view(np.uint64))
    for k, sh in ((1, 2), (2, 3), (3, 4)):
        assert np.array_equal(d.bbox[:, k].view(np.uint64), np.roll(want, -sh).view(np.uint64))
    assert len(d.score) == n


def test_rows_patched_for_the_device_reader(tmp_path):
    """taoamd_pred_patch: the objects the device-side reader (csrc/
    json_ingest.hip) leaves to this one, read at their byte offsets into their
    rows -- the same values as the whole-file read, the same error text."""
    import ctypes as C
    from tao_amodal_amd import columns
    base = '{"image_id": %d, "category_id": 3, "bbox": [1, 2.5, 3e1, 4], "score": %s, "track_id": %s}'
    objs = [base % (k, "0.5", str(k)) for k in range(200)]
    objs[7] = base % (7, "NaN", "7")
    objs[9] = base % (9, "true", "9.0")
    objs[150] = base % (150, "0.1234567890123456789012", "150")
    text = "[" + ",\n ".join(objs) + "]"
    p = str(tmp_path / "p.json")
    with open(p, "w") as f:
        f.write(text)
    whole = DTColumns.from_file_native(p)
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(columns.__file__)),
                              "libtao_amodal_ingest.so"))
    lib.taoamd_pred_patch.argtypes = [C.c_char_p, C.c_int64] + [C.c_void_p] * 8 + [C.c_char_p,
                                                                                   C.c_size_t]
    n = len(objs)
    cols = DTColumns(image_id=np.zeros(n, np.int64), category_id=np.zeros(n, np.int64),
                     bbox=np.zeros((n, 4)), score=np.zeros(n), track_id=np.zeros(n, np.int64),
                     video_id=np.zeros(n, np.int64))
    idx = np.array([7, 9, 150], dtype=np.int64)
    at, pos = [], 0
    for k in range(n):
        pos = text.index("{", pos)
        at.append(pos)
        pos += 1
    at = np.array([at[k] for k in idx], dtype=np.int64)
    err = C.create_string_buffer(512)
    rc = lib.taoamd_pred_patch(os.fsencode(p), 3, idx.ctypes.data, at.ctypes.data,
                               cols.image_id.ctypes.data, cols.category_id.ctypes.data,
                               cols.bbox.ctypes.data, cols.score.ctypes.data,
                               cols.track_id.ctypes.data, cols.video_id.ctypes.data, err, 512)
    assert rc == 0, err.value
    for f in DTColumns.FIELDS:
        a, b = np.asarray(getattr(cols, f))[idx], np.asarray(getattr(whole, f))[idx]
        assert (a.view(np.uint64) == b.view(np.uint64)).all(), f
    # a malformed object: the first one's message, by position in the list
    at_bad = np.array([at[0] + 1, at[1]], dtype=np.int64)
    rc = lib.taoamd_pred_patch(os.fsencode(p), 2, idx[:2].ctypes.data, at_bad.ctypes.data,
                               cols.image_id.ctypes.data, cols.category_id.ctypes.data,
                               cols.bbox.ctypes.data, cols.score.ctypes.data,
                               cols.track_id.ctypes.data, cols.video_id.ctypes.data, err, 512)
    assert rc == 2 and err.value.startswith(b"prediction 7: ")


def test_device_reader_steps_aside_without_a_gpu(tmp_path, monkeypatch):
    """No GPU in the process (this suite): DTColumns.from_file_native reads with
    the host library whatever the file's size."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    monkeypatch.setattr(DTColumns, "DEVICE_INGEST_MIN_BYTES", 0)
    p = str(tmp_path / "p.json")
    with open(p, "w") as f:
        f.write('[{"image_id": 4, "category_id": 2, "bbox": [1, 2, 3, 4], "score": 0.25}]')
    d = DTColumns.from_file_native(p)
    assert len(d) == 1 and d.score[0] == 0.25 and d.track_id[0] == -1
