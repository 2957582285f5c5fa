"""csrc/decfloat.hpp (the decimal -> double conversion of the device-side
prediction reader) against Python's float(), the value json.load hands the
reference (lvis_amodal/results.py:29-30): bit for bit, on the host build of the
same header."""
import ctypes as C
import os
import random
import struct
import subprocess
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim():
    d = tempfile.mkdtemp(prefix="decf")
    so = os.path.join(d, "decf.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared",
                           "-ffp-contract=off", os.path.join(HERE, "decfloat_shim.cpp"),
                           "-o", so])
    lib = C.CDLL(so)
    lib.decf_parse.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_double),
                               C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
    lib.decf_parse_many.argtypes = [C.c_char_p, C.c_void_p, C.c_longlong, C.c_void_p,
                                    C.c_void_p]
    return lib


def many(lib, strings):
    text = ("\n".join(strings) + "\n").encode()
    off = np.zeros(len(strings) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) + 1 for s in strings])
    out = np.zeros(len(strings))
    used = np.zeros(len(strings), dtype=np.int32)
    lib.decf_parse_many(text, off.ctypes.data, len(strings), out.ctypes.data,
                        used.ctypes.data)
    return out, used


def check(lib, strings, may_skip=False):
    out, used = many(lib, strings)
    want = np.array([float(s) for s in strings])
    decided = used > 0
    if not may_skip:
        bad = [s for s, u in zip(strings, used) if u == 0]
        assert not bad, bad[:5]
    assert (used[decided] == np.array([len(s) for s in strings])[decided]).all()
    a = out[decided].view(np.uint64)
    b = want[decided].view(np.uint64)
    if not (a == b).all():
        k = int(np.flatnonzero(a != b)[0])
        s = [s for s, d in zip(strings, decided) if d][k]
        raise AssertionError("%s: got %r want %r" % (s, out[decided][k], want[decided][k]))
    return int(decided.sum())


def test_repr_of_random_doubles(shim):
    rng = random.Random(7)
    strings = []
    for _ in range(400000):
        bits = rng.getrandbits(64)
        x = struct.unpack("<d", struct.pack("<Q", bits))[0]
        if x != x or x in (float("inf"), float("-inf")):
            continue
        s = repr(x)
        strings.append(s)
    # json.dumps / repr spellings: 1e+22, 1.5e-07, 123.456, 0.0001
    check(shim, strings)


def test_scores_and_coordinates(shim):
    rng = np.random.default_rng(3)
    strings = [repr(float(x)) for x in rng.random(300000)]
    strings += [repr(float(x)) for x in rng.uniform(-100, 2000, 200000)]
    strings += ["%.2f" % x for x in rng.uniform(-100, 2000, 100000)]
    strings += ["%d" % x for x in rng.integers(-10 ** 6, 10 ** 6, 100000)]
    check(shim, strings)


def test_digit_strings_and_exponents(shim):
    rng = random.Random(11)
    strings = []
    for _ in range(300000):
        nd = rng.randint(1, 19)
        digits = str(rng.randint(1, 9)) + "".join(rng.choice("0123456789") for _ in range(nd - 1))
        cut = rng.randint(0, nd)
        s = (digits[:cut] or "0") + ("." + digits[cut:] if cut < nd else "")
        if rng.random() < 0.6:
            s += rng.choice("eE") + rng.choice(["", "+", "-"]) + str(rng.randint(0, 340))
        if rng.random() < 0.3:
            s = "-" + s
        if s[0] == "0" and len(s) > 1 and s[1].isdigit():
            continue
        strings.append(s)
    n = check(shim, strings, may_skip=True)
    assert n > 0.99 * len(strings)      # (undecided products are rare)


def test_edges(shim):
    strings = ["0", "-0", "0.0", "-0.0", "0e0", "0.000", "1", "-1", "1e22", "1e23",
               "9007199254740992", "9007199254740993", "9007199254740994",
               "9007199254740995", "4.9e-324", "5e-324", "2.4703282292062327e-324",
               "2.4703282292062328e-324", "2.2250738585072014e-308",
               "2.2250738585072011e-308", "2.225073858507201e-308",
               "1.7976931348623157e308", "1.7976931348623158e308", "1.797693134862316e308",
               "1e308", "1e309", "1e-400", "123456789012345678", "1234567890123456789",
               "0.1", "0.2", "0.30000000000000004", "1e-5", "1.5e-07", "1E5", "1e+5",
               "8.5", "0.5", "0.25", "1000000", "123.456e2", "0.000001e-300"]
    check(shim, strings)
    # powers of two and their neighbours, halfway cases of 17-digit decimals
    more = []
    for e in range(-1070, 1020, 7):
        x = 2.0 ** e
        for y in (x, np.nextafter(x, 0), np.nextafter(x, np.inf)):
            more.append(repr(float(y)))
            more.append("%.17e" % float(y))
            more.append("%.16e" % float(y))
    check(shim, more)


def test_what_is_not_a_json_number(shim):
    out, used = many(shim, ["", "-", "01", "1.", ".5", "+1", "1e", "1e+", "abc", "NaN",
                            "Infinity", "-Infinity", "12345678901234567890", "1e12345"])
    assert (used == 0).all()
    # a number ends where the next byte cannot continue it
    d, i, v = C.c_double(), C.c_int(), C.c_longlong()
    assert shim.decf_parse(b"12.5,", 5, C.byref(d), C.byref(i), C.byref(v)) == 4
    assert d.value == 12.5 and not i.value
    assert shim.decf_parse(b"-42]", 4, C.byref(d), C.byref(i), C.byref(v)) == 3
    assert i.value and v.value == -42 and d.value == -42.0
