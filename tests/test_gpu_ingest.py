"""The prediction file read on the device (csrc/json_ingest.hip) against the
host reader (csrc/ingest.cpp, itself pinned to json.load in tests/
test_ingest.py): the same columns bit for bit, the same errors, and the host
reader taking over wherever the device reader does not decide."""
import json
import os

import numpy as np
import pytest

from tao_amodal_amd.columns import DTColumns

pytestmark = pytest.mark.gpu

FIELDS = DTColumns.FIELDS


@pytest.fixture(autouse=True)
def small_files_too(monkeypatch):
    import torch  # noqa: F401  (the device path needs the runtime torch loaded)
    monkeypatch.setattr(DTColumns, "DEVICE_INGEST_MIN_BYTES", 0)


def host(path):
    os.environ["TAOAMD_DEVICE_INGEST"] = "0"
    try:
        return DTColumns.from_file_native(path)
    finally:
        del os.environ["TAOAMD_DEVICE_INGEST"]


def device(path):
    import ctypes as C
    so = os.path.join(os.path.dirname(os.path.abspath(DTColumns.__module__.replace(".", "/"))), "x")
    from tao_amodal_amd import columns
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(columns.__file__)),
                              "libtao_amodal_ingest.so"))
    return DTColumns._from_file_device(path, lib)


def same(a, b):
    assert len(a) == len(b)
    for f in FIELDS:
        x, y = np.asarray(getattr(a, f)), np.asarray(getattr(b, f))
        assert x.dtype == y.dtype and x.shape == y.shape, f
        assert (x.view(np.uint64) == y.view(np.uint64)).all(), f


def synth_columns(n, seed=0):
    rng = np.random.default_rng(seed)
    return DTColumns(
        image_id=rng.integers(1, 10 ** 6, n), category_id=rng.integers(1, 1204, n),
        bbox=np.concatenate([rng.uniform(-64, 1216, (n, 2)), rng.uniform(1, 400, (n, 2))], 1),
        score=rng.random(n), track_id=rng.integers(0, 10 ** 5, n),
        video_id=rng.integers(0, 2000, n))


def test_native_writer_file_equals_host_reader(tmp_path):
    p = str(tmp_path / "pred.json")
    c = synth_columns(300000, 1)
    c.write_json(p)
    d = device(p)
    assert d is not None
    same(d, host(p))
    same(d, c)


def test_integer_boxes_and_json_dumps_spellings(tmp_path):
    rng = np.random.default_rng(2)
    rows = []
    for k in range(20000):
        r = {"image_id": int(rng.integers(1, 10 ** 9)), "category_id": int(rng.integers(1, 1204)),
             "bbox": [int(x) for x in rng.integers(-50, 1300, 4)] if k % 3 == 0
             else [float(x) for x in rng.uniform(-50, 1300, 4)],
             "score": float(rng.random()) * (1e-7 if k % 11 == 0 else 1.0),
             "track_id": int(rng.integers(0, 10 ** 12)), "video_id": int(rng.integers(0, 3000))}
        if k % 5 == 0:
            del r["track_id"]
        if k % 7 == 0:
            r["extra"] = {"a": [1, 2, {"b": "x}]"}], "c": "str[ing{"}
            r["segmentation"] = [[1.5, 2.5, 3.5]]
        rows.append(r)
    for indent in (None, 1):
        p = str(tmp_path / ("pred%s.json" % indent))
        with open(p, "w") as f:
            json.dump(rows, f, indent=indent)
        d = device(p)
        assert d is not None
        same(d, host(p))


def test_objects_left_to_the_host_reader(tmp_path):
    base = '{"image_id": %d, "category_id": 3, "bbox": [1, 2.5, 3e1, 4], "score": %s, "track_id": %s, "video_id": 1}'
    objs = [base % (k, "0.5", str(k)) for k in range(5000)]
    objs[7] = base % (7, "NaN", "7")
    objs[8] = base % (8, "true", "8")
    objs[9] = base % (9, "0.25", "9.0")                 # an id spelt as a float
    objs[10] = base % (10, "0.1234567890123456789012", "10")      # 22 digits
    objs[11] = base % (11, "-Infinity", "11")
    objs[12] = base % (12, "1e400", "12")
    p = str(tmp_path / "pred.json")
    with open(p, "w") as f:
        f.write("[" + ",\n".join(objs) + "]\n")
    d = device(p)
    assert d is not None
    same(d, host(p))
    assert np.isnan(d.score[7]) and d.score[8] == 1.0 and d.track_id[9] == 9


def test_what_goes_to_the_host_reader_as_a_whole(tmp_path):
    ok = '{"image_id": 1, "category_id": 3, "bbox": [1, 2, 3, 4], "score": 0.5}'
    cases = {
        "backslash": "[" + ok + ', {"image_id": 2, "category_id": 3, "bbox": [1,2,3,4], "score": 1, "n\\u0061me": 3}]',
        "not_a_list": '{"a": ' + ok + "}",
        "number_in_list": "[" + ok + ", 5]",
        "string_in_list": "[" + ok + ', "x"]',
        "nested_list": "[" + ok + ", [" + ok + "]]",
        "text_after": "[" + ok + "] x",
        "two_lists": "[" + ok + "] [" + ok + "]",
        "unterminated": "[" + ok + ", " + ok,
        "empty": "",
    }
    for name, text in cases.items():
        p = str(tmp_path / (name + ".json"))
        with open(p, "w") as f:
            f.write(text)
        assert device(p) is None, name


def test_errors_are_the_host_readers(tmp_path):
    ok = '{"image_id": 1, "category_id": 3, "bbox": [1, 2, 3, 4], "score": 0.5}'
    bad = {
        "missing_key": '{"image_id": 1, "category_id": 3, "bbox": [1, 2, 3, 4]}',
        "short_box": '{"image_id": 1, "category_id": 3, "bbox": [1, 2, 3], "score": 0.5}',
        "garbage": '{"image_id": 1x, "category_id": 3, "bbox": [1, 2, 3, 4], "score": 0.5}',
    }
    for name, obj in bad.items():
        p = str(tmp_path / (name + ".json"))
        with open(p, "w") as f:
            f.write("[" + ", ".join([ok] * 50 + [obj] + [ok] * 50) + "]")
        assert device(p) is None, name            # (the host reader then raises its own error)
        with pytest.raises((KeyError, ValueError)):
            DTColumns.from_file_native(p)


def test_empty_list_and_white_space(tmp_path):
    p = str(tmp_path / "e.json")
    with open(p, "w") as f:
        f.write("  [ \n ]  \n")
    d = device(p)
    assert d is not None and len(d) == 0
    p = str(tmp_path / "w.json")
    with open(p, "w") as f:
        f.write(' [ {"image_id" : 4 ,\n "category_id":\t2, "bbox" : [ 1 , 2 , 3 , 4 ] , "score" : 1 } , ]')
    d = device(p)
    # (a trailing comma: the host reader accepts what stands between objects leniently)
    h = host(p)
    assert d is not None
    same(d, h)


def test_from_file_native_takes_the_device_path(tmp_path):
    p = str(tmp_path / "pred.json")
    c = synth_columns(50000, 5)
    c.write_json(p)
    same(DTColumns.from_file_native(p), c)
    same(DTColumns.from_json(p), c)


def test_columns_stay_on_the_device_and_arrive_on_the_host(tmp_path):
    """DeviceDTColumns: the table builds get the tensors the reader made (no
    upload), every other reader the host arrays; a rebound or edited column
    retires its device copy."""
    import torch
    from tao_amodal_amd import flatten_dev
    from tao_amodal_amd.columns import DeviceDTColumns
    p = str(tmp_path / "pred.json")
    c = synth_columns(80000, 9)
    c.write_json(p)
    d = DTColumns.from_file_native(p)
    assert isinstance(d, DeviceDTColumns) and len(d) == 80000
    raw = flatten_dev.raw_columns(d, "cuda")
    assert raw["score"].is_cuda and raw["area"] is None
    assert raw["score"].data_ptr() == d._dev["score"].data_ptr()       # not a copy
    same(d, c)                                     # (waits for the host arrays)
    assert flatten_dev.raw_columns(d, "cuda")["bbox"].data_ptr() == d._dev["bbox"].data_ptr()
    assert (raw["bbox"].cpu().numpy() == c.bbox).all()
    # an in-place edit of an arrived column is seen; the others keep their copy
    d.score[:] = 0.5
    raw2 = flatten_dev.raw_columns(d, "cuda")
    assert "score" not in d._dev and float(raw2["score"][0]) == 0.5
    # a rebound column, and one handed back as it is
    e = DTColumns.from_file_native(p)
    e.track_id = e.track_id
    assert "track_id" in e._dev
    e.image_id = e.image_id + 1
    assert "image_id" not in e._dev and int(e.image_id[0]) == int(c.image_id[0]) + 1
    raw3 = flatten_dev.raw_columns(e, "cuda")
    assert int(raw3["image_id"][0]) == int(c.image_id[0]) + 1
    t = e.take(np.arange(10))
    assert type(t) is DTColumns and (t.score == c.score[:10]).all()


def test_track_clash_answered_on_the_device(tmp_path):
    from tao_amodal_amd.columns import DeviceDTColumns
    from tao_amodal_amd import flatten
    c = synth_columns(60000, 11)
    c.track_id = c.track_id % 500                   # (every id on ~120 rows)
    c.video_id = c.track_id % 7                     # one video per track id
    p = str(tmp_path / "a.json")
    c.write_json(p)
    d = DTColumns.from_file_native(p)
    assert isinstance(d, DeviceDTColumns) and d.track_clash_free() is True
    assert flatten.make_track_ids_unique(d)[1] == 0
    c.video_id = c.video_id.copy()
    c.video_id[123] += 1                            # that id now has two videos
    p = str(tmp_path / "b.json")
    c.write_json(p)
    d = DTColumns.from_file_native(p)
    assert d.track_clash_free() is False
    assert flatten.make_track_ids_unique(d)[1] == 1
    d.track_id = d.track_id.copy()                  # a replaced column: the host decides
    assert d.track_clash_free() is None


def test_random_files_agree_with_the_host_reader_or_go_to_it(tmp_path):
    """Fuzz: prediction lists with random spellings, spacing, key order, extra
    and missing keys, literals and broken syntax.  The device reader either
    steps aside (None) or returns exactly what the host reader returns; where
    the host reader raises, the device reader must have stepped aside."""
    import random
    rng = random.Random(20240807)

    def number(kind):
        r = rng.random()
        if kind == "id":
            if r < 0.85:
                return str(rng.randint(0, 10 ** rng.randint(1, 17)))
            if r < 0.9:
                return "-" + str(rng.randint(0, 1000))
            if r < 0.95:
                return "%d.0" % rng.randint(0, 1000)
            return rng.choice(["1e3", "12.5", "007", "true", "null"])
        if r < 0.5:
            return repr(rng.uniform(-100, 2000))
        if r < 0.65:
            return str(rng.randint(-50, 2000))
        if r < 0.8:
            return "%.*f" % (rng.randint(1, 6), rng.uniform(0, 1))
        if r < 0.9:
            return repr(rng.uniform(0, 1) * 10.0 ** rng.randint(-12, 12))
        return rng.choice(["NaN", "Infinity", "-Infinity", "true", "false", "null", "1e400",
                           "0.12345678901234567890123", "1.", ".5", "1e", "0x10", "-0.0", "0",
                           "-0", "5e-324", "1E+2"])

    def ws():
        return rng.choice(["", "", "", " ", "\n", "\t", "  ", " \r\n "])

    def obj(plain=False):
        fields = [("image_id", number("id")), ("category_id", number("id")),
                  ("score", number("f")),
                  ("bbox", "[" + ws() + ("," + ws()).join(number("f") for _ in range(
                      4 if plain else rng.choice([4, 4, 4, 4, 4, 4, 3, 5, 0]))) + ws() + "]")]
        if rng.random() < 0.7:
            fields.append(("track_id", number("id")))
        if rng.random() < 0.7:
            fields.append(("video_id", number("id")))
        if rng.random() < 0.2:
            fields.append((rng.choice(["extra", "segmentation", "area", "name"]),
                           rng.choice(['"text"', '{"a": [1, {"b": "}]"}], "c": null}', "[[1, 2], []]",
                                       "12", "null", '"br{ace[s"', "true"])))
        if rng.random() < 0.05 and not plain:
            fields.pop(rng.randrange(len(fields)))               # a missing key
        if rng.random() < 0.05:
            fields.append(rng.choice(fields))                    # a key twice: the last one wins
        rng.shuffle(fields)
        sep = rng.choice([",", ", ", " ,\n"]) if plain or rng.random() < 0.98 else " "
        body = sep.join('"%s"%s:%s%s' % (k, ws(), ws(), v) for k, v in fields)
        return "{" + ws() + body + ws() + "}"

    agreed = stepped_aside = 0
    for case in range(300):
        n = rng.choice([0, 1, 2, 5, 40, 300])
        # (most files hold only plain objects: the device reader's own path)
        plain = rng.random() < 0.6
        if plain:
            st = rng.getstate()
        objs = []
        for _ in range(n):
            o = obj(plain)
            if plain:
                # keep drawing until the host reader would accept the object as plain
                for _try in range(20):
                    if not any(t in o for t in ("NaN", "Infinity", "true", "false", "null", "e400",
                                                "1.,", ".5", "1e,", "0x", "007", "8901234567890123",
                                                '1e"', "1e ", "1e]", "1e}")) \
                            and all(k in o for k in ('"image_id"', '"category_id"', '"score"',
                                                     '"bbox"')):
                        break
                    o = obj(plain)
            objs.append(o)
        text = ws() + "[" + ws() + ("," + ws()).join(objs) + ws() + "]" + ws()
        if rng.random() < 0.03:
            text = text.replace("]", "", 1)
        p = str(tmp_path / ("f%d.json" % case))
        with open(p, "w") as f:
            f.write(text)
        try:
            want = host(p)
        except Exception:
            want = None
        got = device(p)
        if want is None:
            assert got is None, (case, text[:300])
            stepped_aside += 1
            continue
        if got is None:
            stepped_aside += 1
            continue
        same(got, want)
        agreed += 1
    assert agreed >= 100 and stepped_aside >= 20, (agreed, stepped_aside)
