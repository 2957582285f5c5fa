"""numpy restatement of the chunk format of csrc/exchange.hip (test
infrastructure: the gloo tests use it as the backend, the GPU tests compare
the HIP kernels with it byte for byte)."""
import numpy as np

N_THR, N_REC = 10, 101
REC_THRS = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1,
                       endpoint=True)


def align(x):
    return (x + 255) & ~255


def layout(block_cats, n_rng, capacity):
    rows = block_cats * n_rng
    hdr = align(rows * 8)          # num_gt[rows], off[rows]
    rec = align(rows * N_THR * 8)
    return hdr, rec, hdr + rec + align(capacity * 8)


def run_map(ng):
    """Run index of each recall column for a row with ng ground truths."""
    rc = np.arange(ng + 1, dtype=np.float64) / ng
    c = np.searchsorted(rc, REC_THRS, side="left")
    d = np.zeros(N_REC, np.int64)
    d[1:] = np.cumsum(c[1:] != c[:-1])
    return d


def sizes(block_cats, n_rng, world, num_gt):
    ng = np.asarray(num_gt).reshape(world, block_cats * n_rng)
    out = np.zeros(world, np.int64)
    for b in range(world):
        for n in ng[b]:
            if n > 0:
                out[b] += (run_map(int(n))[-1] + 1) * N_THR
    return out


def pack(n_cat, n_rng, block_cats, rank, num_gt, val, rec, capacity):
    """-> chunk bytes (uint8) of `rank`; tables addressed by global row."""
    hdr, recb, total = layout(block_cats, n_rng, capacity)
    rows = block_cats * n_rng
    chunk = np.zeros(total, np.uint8)
    h = chunk[:rows * 4].view(np.int32)
    ho = chunk[rows * 4:rows * 8].view(np.int32)
    r = chunk[hdr:hdr + rows * N_THR * 8].view(np.float64).reshape(rows, N_THR)
    lv = chunk[hdr + recb:hdr + recb + capacity * 8].view(np.float64)
    ng = np.asarray(num_gt).reshape(-1)
    v = np.asarray(val).reshape(-1, N_THR, N_REC)
    rr = np.asarray(rec).reshape(-1, N_THR)
    off = 0
    r[:] = -1.0
    for i in range(rows):
        row = rank * rows + i
        n = int(ng[row]) if row < n_cat * n_rng else 0
        h[i] = n
        if n <= 0:
            continue
        r[i] = rr[row]
        ho[i] = off
        d = run_map(n)
        first = np.r_[0, np.flatnonzero(d[1:] != d[:-1]) + 1]
        nd = len(first)
        lv[off:off + nd * N_THR] = v[row][:, first].reshape(-1)
        off += nd * N_THR
    return chunk


def decode_records(bits):
    """(tp << 32 | n) records of the device sweeps -> tp / (n + eps), the
    reference's precision expression (lvis_amodal/eval.py:384)."""
    bits = np.asarray(bits).view(np.uint64)
    tp = (bits >> np.uint64(32)).astype(np.float64)
    n = (bits & np.uint64(0xffffffff)).astype(np.float64)
    return tp / (n + np.spacing(1))


def unpack(n_cat, n_rng, block_cats, world, chunks, capacity, records=False):
    """records=True: the levels are the device sweeps' (tp, n) records."""
    hdr, recb, total = layout(block_cats, n_rng, capacity)
    rows = block_cats * n_rng
    chunks = np.asarray(chunks).view(np.uint8)
    KR = n_cat * n_rng
    precision = np.full((N_THR, N_REC, KR), -1.0)
    recall = np.full((N_THR, KR), -1.0)
    num_gt = np.zeros(KR, np.int32)
    for b in range(world):
        c = chunks[b * total:(b + 1) * total]
        h = c[:rows * 4].view(np.int32)
        ho = c[rows * 4:rows * 8].view(np.int32)
        r = c[hdr:hdr + rows * N_THR * 8].view(np.float64).reshape(rows, N_THR)
        lv = c[hdr + recb:hdr + recb + capacity * 8].view(np.float64)
        if records:
            lv = decode_records(lv)
        for i in range(rows):
            row = b * rows + i
            if row >= KR:
                break
            n = int(h[i])
            num_gt[row] = n
            recall[:, row] = r[i]
            if n <= 0:
                continue
            d = run_map(n)
            nd = d[-1] + 1
            off = int(ho[i])
            block = lv[off:off + nd * N_THR].reshape(N_THR, nd)
            precision[:, :, row] = block[:, d]
    return (num_gt.reshape(n_cat, n_rng),
            precision.reshape(N_THR, N_REC, n_cat, n_rng),
            recall.reshape(N_THR, n_cat, n_rng))
