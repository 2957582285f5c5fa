"""The C-ABI library loads on a CPU-only host and exports every symbol that
include/tao_amodal_hip.h declares (no compute calls without a GPU)."""
import os
import re

import numpy as np

from tao_amodal_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    text = open(os.path.join(ROOT, "include", "tao_amodal_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(taoamd_[a-z_0-9]+)\s*\(", text))
    assert declared, "no declarations found"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_ingest_header_symbols_are_exported():
    import ctypes
    text = open(os.path.join(ROOT, "include", "tao_amodal_ingest.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(taoamd_[a-z_0-9]+)\s*\(", text))
    assert len(declared) == 37, declared
    lib = ctypes.CDLL(os.path.join(ROOT, "tao_amodal_amd", "libtao_amodal_ingest.so"))
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_thresholds_are_numpy_linspace_bit_for_bit():
    lib = _lib.load()
    a, b = np.zeros(10), np.zeros(101)
    assert lib.taoamd_thresholds_host(a.ctypes.data, b.ctypes.data) == 0
    assert np.array_equal(a, np.linspace(0.5, 0.95, 10))
    assert np.array_equal(b, np.linspace(0.0, 1.0, 101))


def test_status_strings():
    lib = _lib.load()
    assert lib.taoamd_strerror(0) == b"ok"
    assert b"workspace" in lib.taoamd_strerror(4)
    assert lib.taoamd_version() >= 100


def test_sort_plan_covers_every_category():
    """taoamd_sort_plan_host (host only): the chunks tile every category, a
    chunk's buckets and scatter tiles are consecutive, sizes agree."""
    from tao_amodal_amd import engine
    chunk, tile = 16 * _lib.SEGMENT_TILE, _lib.SEGMENT_TILE
    sizes = [0, 5, 1024, 1025, 2817, chunk, chunk + 1, 0, 2 * chunk + 3000, 352 * 3 + 1]
    cat_off = np.zeros(len(sizes) + 1, np.int32)
    np.cumsum(sizes, out=cat_off[1:])
    (chunks, split, stile, bucket), (nc, ns, nt, nb), merge = engine.sort_plan(cat_off)
    assert merge and len(chunks) == nc and len(bucket) == nb and len(stile) == nt
    at, b0, t0, splits = {}, 0, 0, []
    for i, c in enumerate(chunks):
        begin, n, bucket0, nbk, stile0, final, cat = (int(x) for x in c[:7])
        assert bucket0 == b0 and 0 < n <= chunk
        assert begin == at.get(cat, int(cat_off[cat]))
        at[cat] = begin + n
        assert nbk == (1 if n <= 1024 else -(-n // 352)) and nbk <= 128
        assert final == (sizes[cat] <= chunk)
        assert (bucket[b0:b0 + nbk] == i).all()
        if nbk > 1:
            splits.append(i)
            assert stile0 == t0
            k = -(-n // tile)
            assert (stile[t0:t0 + k] == i).all()
            t0 += k
        b0 += nbk
    assert split[:ns].tolist() == splits and b0 == nb and t0 == nt
    assert all(at.get(k, int(cat_off[k])) == cat_off[k + 1] for k in range(len(sizes)))


def test_all_in_sorted_is_the_membership_test():
    import ctypes as C
    from tao_amodal_amd.columns import _ingest_lib
    lib = _ingest_lib()
    lib.taoamd_host_all_in_sorted.restype = C.c_int
    lib.taoamd_host_all_in_sorted.argtypes = [C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    rng = np.random.default_rng(3)

    def ask(keys, values):
        k = np.ascontiguousarray(keys, dtype=np.int64)
        v = np.ascontiguousarray(values, dtype=np.int64)
        return lib.taoamd_host_all_in_sorted(len(k), k.ctypes.data, len(v), v.ctypes.data)
    for span in (50, 1 << 20, 1 << 40):            # bitmap and binary-search branches
        keys = np.unique(rng.integers(-span, span, 2000))
        inside = rng.choice(keys, 100000)
        assert ask(keys, inside) == 1
        outside = inside.copy()
        missing = np.setdiff1d(np.arange(keys[0] - 2, keys[-1] + 3), keys)[:1] if span == 50 \
            else np.array([keys[-1] + 1])
        outside[77777] = missing[0]
        assert ask(keys, outside) == 0
    assert ask([], []) == 1 and ask([1], []) == 1 and ask([], [1]) == 0


def test_dense_only_entry_points_refuse_the_pair_layout():
    """`ignored == matched + 1` selects the interleaved (matched,
    ignored) pair layout in taoamd_match / taoamd_accumulate*; the entry points
    that only know dense tables must refuse that relation (before touching the
    device: this runs without a GPU) instead of scrambling rows."""
    import numpy as np
    from tao_amodal_amd import _lib
    lib = _lib.load()
    buf = np.zeros(64, dtype=np.uint64)
    base = buf.ctypes.data
    order = np.zeros(4, dtype=np.int32)
    # taoamd_gather_rows(n, n_words, src_m, src_i, stride, order, dst_m, dst_i, stream)
    assert lib.taoamd_gather_rows(4, 1, base, base + 32, 1, order.ctypes.data,
                                  base + 256, base + 256 + 8, None) == 2
    # taoamd_exchange_place(n_recv, world, n_words, rows, own_rows, own_rank, src_base,
    #                       pos, out, stream): tables of 16-byte pairs only
    i64 = np.zeros(16, dtype=np.int64)
    p = i64.ctypes.data
    assert lib.taoamd_exchange_place(4, 1, 1, base + 8, base + 8, 0, p, p, base + 256,
                                     None) == 2
