// CPython's set of ints, restated for one purpose: the order in which
// ``set(gt_track.keys()) | set(dt_track.keys())`` is iterated -- the order the
// reference adds a track pair's frames in (tao_amodal/eval.py:57,83,109).
//
// Follows Objects/setobject.c of CPython 3.7 - 3.12 (one algorithm: open
// addressing, LINEAR_PROBES = 9 neighbours per probe, then the perturbed
// recurrence i = 5 i + 1 + perturb; growth at fill * 5 >= mask * 3):
//
//   set(iterable)   set_add_entry() key by key, table quadrupled on the way
//   a | b           set_or(): set_copy(a) -- set_merge() into an empty set: one
//                   resize to (used * 2), then a slot-for-slot copy when the
//                   sizes agree, else set_insert_clean() in a's slot order --
//                   then set_merge(result, b): one resize ahead when
//                   (fill + b.used) * 5 >= mask * 3, then set_add_entry() of
//                   b's keys in b's slot order
//   iteration       ascending slot
//
// Keys are non-negative ints below 2^61 - 1 (hash(k) == k); nothing is ever
// deleted, so there are no dummy entries (fill == used).  A table holds int32
// CODES (0 = unused slot); what a code's key hashes to comes from the caller's
// functor.  Pinned against the interpreter's own sets in tests/test_pyset.py.
//
// Host + device: the same text runs in the guard kernel (a thread per listed
// pair) and behind taoamd_*_host entry points for the CPU tests.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define PYSET_HD __host__ __device__ __forceinline__
#else
#define PYSET_HD inline
#endif

namespace pyset {

constexpr int LINEAR_PROBES = 9;
constexpr int PERTURB_SHIFT = 5;
constexpr uint32_t MINSIZE = 8;

struct Set {
    int32_t *t;      // mask + 1 slots
    uint32_t mask;
    uint32_t used;
};

// smallest power of two > minused, at least MINSIZE (set_table_resize)
PYSET_HD uint32_t table_size(uint64_t minused)
{
    uint64_t n = MINSIZE;
    while (n <= minused) n <<= 1;
    return (uint32_t)n;
}

// (a set of n keys never has more than table_size(4 n) slots: a resize asks
// for at most 4 * used -- callers size their buffers for the union's n)

PYSET_HD void clear(int32_t *t, uint32_t size)
{
    for (uint32_t i = 0; i < size; i++) t[i] = 0;
}

// set_insert_clean(): the key is known to be absent, no dummies
PYSET_HD void insert_clean(int32_t *t, uint32_t mask, int32_t code, uint64_t hash)
{
    uint64_t perturb = hash;
    uint64_t i = hash & mask;
    for (;;) {
        if (t[i] == 0) { t[i] = code; return; }
        if (i + LINEAR_PROBES <= mask) {
            for (int j = 1; j <= LINEAR_PROBES; j++)
                if (t[i + j] == 0) { t[i + j] = code; return; }
        }
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & mask;
    }
}

// set_table_resize(): the entries re-inserted in slot order into `spare`
// (which becomes the table; the old table becomes the spare)
template <class Hash>
PYSET_HD void resize(Set &s, uint64_t minused, int32_t *&spare, const Hash &hash)
{
    const uint32_t size = table_size(minused);
    clear(spare, size);
    for (uint32_t i = 0; i <= s.mask; i++) {
        const int32_t c = s.t[i];
        if (c) insert_clean(spare, size - 1, c, hash(c));
    }
    int32_t *old = s.t;
    s.t = spare;
    s.mask = size - 1;
    spare = old;
}

// set_add_entry(); two codes are the same key iff they are equal
template <class Hash>
PYSET_HD void add(Set &s, int32_t code, int32_t *&spare, const Hash &hash)
{
    const uint64_t h = hash(code);
    const uint64_t mask = s.mask;
    uint64_t perturb = h;
    uint64_t i = h & mask;
    for (;;) {
        uint64_t e = i;
        int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
        do {
            const int32_t c = s.t[e];
            if (c == 0) goto found_unused;
            if (c == code) return;
            e++;
        } while (probes--);
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & mask;
        continue;
    found_unused:
        s.t[e] = code;
        s.used++;
        if ((uint64_t)s.used * 5 < mask * 3) return;
        resize(s, s.used > 50000 ? (uint64_t)s.used * 2 : (uint64_t)s.used * 4, spare, hash);
        return;
    }
}

// ``set(a) | set(b)``: a's and b's codes come from the functors code_a(k),
// code_b(k) in insertion order.  `buf` = three buffers of cap >=
// table_size(4 (na + nb)) slots each.  Returns the union's table (one of the buffers).
template <class CodeA, class CodeB, class Hash>
PYSET_HD Set set_union(uint32_t na, const CodeA &code_a, uint32_t nb, const CodeB &code_b,
                       const Hash &hash, int32_t *buf, uint32_t cap)
{
    int32_t *b0 = buf, *b1 = buf + cap, *b2 = buf + 2 * (uint64_t)cap;
    // B = set(b): ping-pong between b0 / b1
    Set B{b0, MINSIZE - 1, 0};
    int32_t *spare = b1;
    clear(B.t, MINSIZE);
    for (uint32_t k = 0; k < nb; k++) add(B, code_b(k), spare, hash);
    // A = set(a): ping-pong between the free one and b2
    Set A{spare, MINSIZE - 1, 0};
    spare = b2;
    clear(A.t, MINSIZE);
    for (uint32_t k = 0; k < na; k++) add(A, code_a(k), spare, hash);
    // R = set_copy(A) = set_merge(empty, A)
    Set R;
    if (A.used == 0) {
        R = Set{spare, MINSIZE - 1, 0};
        clear(R.t, MINSIZE);
        spare = A.t;
    } else {
        uint32_t rsize = MINSIZE;
        if ((uint64_t)A.used * 5 >= (uint64_t)(MINSIZE - 1) * 3) rsize = table_size(2ull * A.used);
        if (rsize - 1 == A.mask) {
            R = A;                          // the slot-for-slot copy: A itself
        } else {
            clear(spare, rsize);
            for (uint32_t i = 0; i <= A.mask; i++) {
                const int32_t c = A.t[i];
                if (c) insert_clean(spare, rsize - 1, c, hash(c));
            }
            R = Set{spare, rsize - 1, A.used};
            spare = A.t;
        }
    }
    // set_merge(R, B)
    if (B.used) {
        if ((uint64_t)(R.used + B.used) * 5 >= (uint64_t)R.mask * 3)
            resize(R, 2ull * (R.used + B.used), spare, hash);
        if (R.used == 0 && R.mask == B.mask) {
            return B;                       // (empty target, equal sizes: copied as is)
        }
        if (R.used == 0) {
            for (uint32_t i = 0; i <= B.mask; i++) {
                const int32_t c = B.t[i];
                if (c) insert_clean(R.t, R.mask, c, hash(c));
            }
            R.used = B.used;
            return R;
        }
        for (uint32_t i = 0; i <= B.mask; i++) {
            const int32_t c = B.t[i];
            if (c) add(R, c, spare, hash);
        }
    }
    return R;
}

// numpy's pairwise summation of np.add.reduce over a contiguous float64
// array (blocks of <= 128 with eight running sums, halves above that), fed by
// a generator: the recursion reads every element once, in ascending order
template <class Next>
PYSET_HD double pairwise_block(Next &next, uint32_t n)
{
    if (n < 8) {
        double r = 0.0;                     // (numpy: -0.0 start only matters for -0.0 terms)
        for (uint32_t i = 0; i < n; i++) r += next();
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = next();
    uint32_t i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += next();
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += next();
    return res;
}

// the halving above 128 elements, without recursion: an explicit stack of
// (length, phase) frames; depth <= 32
template <class Next>
PYSET_HD double pairwise_sum(Next &next, uint32_t n)
{
    uint32_t len[34];
    double left[34];
    uint8_t phase[34];
    int sp = 0;
    len[0] = n;
    phase[0] = 0;
    double ret = 0.0;
    while (sp >= 0) {
        const uint32_t m = len[sp];
        if (m <= 128) {
            ret = pairwise_block(next, m);
            sp--;
            continue;
        }
        uint32_t n2 = m / 2;
        n2 -= n2 % 8;
        if (phase[sp] == 0) {
            phase[sp] = 1;
            sp++;
            len[sp] = n2;
            phase[sp] = 0;
        } else if (phase[sp] == 1) {
            left[sp] = ret;
            phase[sp] = 2;
            sp++;
            len[sp] = m - n2;
            phase[sp] = 0;
        } else {
            ret = left[sp] + ret;
            sp--;
        }
    }
    return ret;
}

// One track's frames: timeline positions ascending, one box (x, y, w, h) each.
struct Frames {
    const int32_t *pos;
    const double *box;
    int32_t n;
};

PYSET_HD int32_t find_pos(const Frames &f, int32_t p)
{
    int32_t lo = 0, hi = f.n;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (f.pos[mid] < p) lo = mid + 1; else hi = mid;
    }
    return (lo < f.n && f.pos[lo] == p) ? lo : -1;
}

// 3D IoU (mode 0, compute_track_box_iou, T/eval.py:73-96) or average IoU
// (mode 1, compute_avg_track_iou, T/eval.py:99-117) of one (detection track,
// GT track) pair with the frames visited in the reference's order.
// tl_id[p] = image id at timeline position p of the pair's video; the dict
// keys of a track are its frames in ascending position.  Per-frame arithmetic
// = bb_intersect_union (T/eval.py:32-48).  Compile with -ffp-contract=off.
PYSET_HD double set_order_iou(const int64_t *tl_id, const Frames &d, const Frames &g,
                              int mode, int32_t *buf, uint32_t cap)
{
    auto hash = [tl_id](int32_t code) { return (uint64_t)tl_id[code - 1]; };
    auto code_g = [&g](uint32_t k) { return g.pos[k] + 1; };
    auto code_d = [&d](uint32_t k) { return d.pos[k] + 1; };
    const Set R = set_union((uint32_t)g.n, code_g, (uint32_t)d.n, code_d, hash, buf, cap);
    uint32_t slot = 0;
    double i = 0.0, u = 0.0;
    // the next key's (intersection, union) terms; ratio for mode 1
    auto next_terms = [&](double &i_, double &u_) -> bool {
        while (slot <= R.mask && R.t[slot] == 0) slot++;
        if (slot > R.mask) return false;
        const int32_t p = R.t[slot++] - 1;
        const int32_t kd = find_pos(d, p), kg = find_pos(g, p);
        if (kd >= 0 && kg >= 0) {
            const double *D = d.box + 4 * (int64_t)kd, *G = g.box + 4 * (int64_t)kg;
            const double da = D[2] * D[3], ga = G[2] * G[3];
            // (Python's max(a, b) is b only if b > a, min(a, b) is b only if b < a)
            const double left = G[0] > D[0] ? G[0] : D[0];
            const double r1 = D[0] + D[2], r2 = G[0] + G[2];
            const double right = r2 < r1 ? r2 : r1;
            const double top = G[1] > D[1] ? G[1] : D[1];
            const double b1 = D[1] + D[3], b2 = G[1] + G[3];
            const double bottom = b2 < b1 ? b2 : b1;
            double w = right - left, h = bottom - top;
            w = 0.0 > w ? 0.0 : w;
            h = 0.0 > h ? 0.0 : h;
            i_ = w * h;
            u_ = da + ga - i_;
            return true;
        }
        const double *B = kd >= 0 ? d.box + 4 * (int64_t)kd : g.box + 4 * (int64_t)kg;
        i_ = -1.0;                          // one side only
        u_ = B[2] * B[3];
        return true;
    };
    if (mode == 0) {
        double i_, u_;
        while (next_terms(i_, u_)) {
            if (i_ >= 0) i += i_;
            u += u_;
        }
        return u > 0 ? i / u : 0.0;
    }
    auto next_ratio = [&]() -> double {
        double i_ = 0.0, u_ = 0.0;
        next_terms(i_, u_);
        return i_ >= 0 ? (u_ > 0 ? i_ / u_ : 0.0) : 0.0;
    };
    return pairwise_sum(next_ratio, R.used) / (double)R.used;
}

}  // namespace pyset
