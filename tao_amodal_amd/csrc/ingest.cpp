// Native columnar ingest of the two input files (host only, no GPU code):
// the prediction list (taoamd_pred_*) and the annotation file (taoamd_gt_*).
//
// The reference loads `prediction.json` with json.load into ~N Python dicts
// (twice: reference lvis_amodal/results.py:29-30 and tools/eval_on_tao_amodal.py:
// 127-128) and keeps working on dicts.  Here the file is parsed straight into
// the six columns of `DTColumns` (image_id, category_id, bbox[4], score,
// track_id, video_id):
//
//   pass 1  one scan finds the byte range of every top-level object
//           (string/escape aware depth counter)
//   pass 2  OpenMP over objects: a small recursive-descent reader pulls the
//           known keys, skips everything else (any JSON value)
//
// The annotation file ({info, images, videos, tracks, annotations,
// categories}; reference tao_amodal/evaluation/tao_amodal/tao.py:4-60,108-160)
// goes the same way into the arrays of `GTColumns`: the top-level object is
// walked once, each of the five tables is cut into objects and read in
// parallel, ragged id lists become CSR.
//
// Numbers are converted with std::from_chars (correctly rounded, the same
// value Python's float() gives), so the columns are bit-identical to what
// `json.load` + numpy conversion produce (tests/test_ingest.py).
#include "../../include/tao_amodal_ingest.h"
#include "host_threads.hpp"

#include <omp.h>
#include <parallel/algorithm>

#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

// std::vector whose resize() leaves trivial elements uninitialised: a vector of
// 30 M element ranges or annotation columns is hundreds of MB, and value-
// initialising it is ONE thread touching every page first (the page faults of
// the big arrays were most of the parse time at 30 M predictions); the
// parallel loops that fill the arrays fault them in instead.
template <class T>
struct NoInit {
    typedef T value_type;
    NoInit() = default;
    template <class U> NoInit(const NoInit<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(::operator new(n * sizeof(T))); }
    void deallocate(T *p, size_t) { ::operator delete(p); }
    template <class U> void construct(U *p) { ::new ((void *)p) U; }       // default-init: nothing
    template <class U, class... A> void construct(U *p, A &&... a) { ::new ((void *)p) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const NoInit<U> &) const { return true; }
    template <class U> bool operator!=(const NoInit<U> &) const { return false; }
};
template <class T> using Vec = std::vector<T, NoInit<T>>;

// byte range [first, second) of one element of a JSON list
struct Span {
    const char *first, *second;
};
typedef Vec<Span> Ranges;

struct Columns {
    std::vector<int64_t> image_id, category_id, track_id, video_id;
    std::vector<double> bbox, score;
    std::string error;
    int64_t first = 0, total = 0;     // a share of the list (taoamd_pred_parse_part)
};

struct Cursor {
    const char *p, *e;
    bool fail = false;
    std::string why;

    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool eat(char c) { ws(); if (p < e && *p == c) { p++; return true; } return false; }
    void bad(const char *msg) { if (!fail) { fail = true; why = msg; } }

    // string without unescaping; returns [b, e) of the raw contents
    bool str(const char *&b, const char *&en)
    {
        ws();
        if (p >= e || *p != '"') { bad("expected string"); return false; }
        b = ++p;
        while (p < e && *p != '"') { if (*p == '\\') p++; p++; }
        if (p >= e) { bad("unterminated string"); return false; }
        en = p++;
        return true;
    }
    bool number(double &v)
    {
        ws();
        const char *s = p;
        {
            // The common case in one pass: [-]digits[.digits], at most 15 digits
            // in all and no exponent.  The digits are an integer m < 10^15 < 2^53
            // and 10^k (k <= 22) is a double too, so m / 10^k is ONE correctly
            // rounded IEEE operation on exact operands: the double strtod /
            // Python's float() give (Clinger's fast path).  Everything else --
            // 16-17 digit scores, exponents, NaN -- takes the general route.
            static const double P10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7,
                                           1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
            const char *q = s;
            const bool neg = q < e && *q == '-';
            if (neg) q++;
            uint64_t m = 0;
            int nd = 0, frac = 0;
            while (q < e && (unsigned)(*q - '0') <= 9u && nd < 16) { m = m * 10 + (uint64_t)(*q - '0'); q++; nd++; }
            bool ok = nd > 0 && nd <= 15;
            if (ok && q < e && *q == '.') {
                q++;
                while (q < e && (unsigned)(*q - '0') <= 9u && nd < 16) {
                    m = m * 10 + (uint64_t)(*q - '0'); q++; nd++; frac++;
                }
                ok = frac > 0 && nd <= 15;
            }
            if (ok && q < e && (*q == ',' || *q == ']' || *q == '}' || *q == ' ' || *q == '\n' ||
                                *q == '\t' || *q == '\r')) {
                const double x = frac ? (double)m / P10[frac] : (double)m;
                v = neg ? -x : x;
                p = q;
                return true;
            }
        }
        if (p < e && *p == '-') p++;
        while (p < e && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' ||
                         *p == 'E' || *p == '+' || *p == '-')) p++;
        if (p == s) {   // NaN / Infinity / -Infinity as json.load accepts them
            if (e - p >= 3 && !strncmp(p, "NaN", 3)) { p += 3; v = NAN; return true; }
            if (e - p >= 8 && !strncmp(p, "Infinity", 8)) { p += 8; v = INFINITY; return true; }
            bad("expected number"); return false;
        }
        if (p - s == 1 && *s == '-' && e - p >= 8 && !strncmp(p, "Infinity", 8)) {
            p += 8; v = -INFINITY; return true;
        }
        auto r = std::from_chars(s, p, v);
        if (r.ec == std::errc::result_out_of_range && r.ptr == p) {
            // 1e400 -> inf, 1e-400 -> 0.0, as json.load (float()) gives them
            v = strtod(std::string(s, p).c_str(), nullptr);
            return true;
        }
        if (r.ec != std::errc() || r.ptr != p) { bad("malformed number"); return false; }
        return true;
    }
    // integer-valued ids keep all 64 bits (a double would lose ids > 2^53)
    bool integer(int64_t &v)
    {
        ws();
        const char *s = p;
        const char *q = p;
        if (q < e && *q == '-') q++;
        while (q < e && *q >= '0' && *q <= '9') q++;
        if (q > s && (q >= e || (*q != '.' && *q != 'e' && *q != 'E'))) {
            auto r = std::from_chars(s, q, v);
            if (r.ec == std::errc() && r.ptr == q) { p = q; return true; }
        }
        double d;
        if (!number(d)) return false;
        v = (int64_t)d;
        return true;
    }
    void skip()   // any JSON value
    {
        ws();
        if (p >= e) { bad("unexpected end"); return; }
        char c = *p;
        if (c == '"') { const char *a, *b; str(a, b); }
        else if (c == '{' || c == '[') {
            int depth = 0;
            while (p < e) {
                char d = *p;
                if (d == '"') { const char *a, *b; str(a, b); continue; }
                if (d == '{' || d == '[') depth++;
                else if (d == '}' || d == ']') { depth--; if (depth == 0) { p++; return; } }
                p++;
            }
            bad("unterminated container");
        } else {
            while (p < e && *p != ',' && *p != '}' && *p != ']' && *p != ' ' &&
                   *p != '\n' && *p != '\t' && *p != '\r') p++;
        }
    }
};

// Large inputs are scanned in parallel, simdjson-style, in two sweeps over
// per-thread chunks.  (A) unescaped quotes per chunk -> is a chunk's first
// byte inside a string (a chunk without a backslash, i.e. every chunk of a
// typical prediction file, is a plain vectorised count).  (B) one bracket
// walk per chunk that does not know the depth `base` its chunk starts at: it
// tracks the depth relative to the chunk start and its running minimum m.
// The absolute depth inside the list is never negative, so base >= -m at any
// time, and an element boundary (absolute depth 0) can only be an event AT the
// running minimum: opens at relative depth m are the candidate starts, closes
// that return to m -- or lower it -- the candidate ends.  Every lowering of m
// opens a new segment of candidates; once the serial prefix over the chunks'
// depth changes has given `base`, the segment of level -base is the chunk's
// true boundary list and the close that opened the segment of level -base-1
// (if any) is the list's closing bracket.  Chunk starts are moved past
// backslash runs, so an escape never straddles two chunks.
struct ChunkWalk {
    struct Seg { int64_t level; size_t s0, e0; const char *opened_by; };
    std::vector<const char *> starts, ends;
    std::vector<Seg> segs;
    int64_t delta = 0, lowest = 0;

    int64_t d = 0, m = 0;       // depth relative to the chunk start, its running minimum

    inline void open_at(const char *p) { if (d == m) starts.push_back(p); d++; }
    inline void close_at(const char *p)
    {
        d--;
        if (d < m) {
            m = d;
            segs.push_back(Seg{m, starts.size(), ends.size(), p});
            ends.push_back(p + 1);
        } else if (d == m) ends.push_back(p + 1);
    }
    void bytes(const char *b, const char *e, bool s)
    {
        for (const char *p = b; p < e; p++) {
            const char c = *p;
            if (s) { if (c == '\\') p++; else if (c == '"') s = false; continue; }
            if (c == '"') s = true;
            else if (c == '{' || c == '[') open_at(p);
            else if (c == '}' || c == ']') close_at(p);
        }
    }
    // A chunk WITHOUT a backslash (every chunk of an ordinary prediction file),
    // 64 bytes at a time: bit masks of the quotes and of the brackets, the
    // in-string mask as the running parity of the quotes, and only the brackets
    // outside strings are visited (~4 of the ~120 bytes of a prediction).
    __attribute__((target("avx2"))) void blocks(const char *b, const char *e, bool s)
    {
        const __m256i quote = _mm256_set1_epi8('"'), fold = _mm256_set1_epi8(0x20);
        const __m256i opn = _mm256_set1_epi8(0x7b), cls = _mm256_set1_epi8(0x7d);
        uint64_t inside = s ? ~0ull : 0ull;
        const char *p = b;
        for (; p + 64 <= e; p += 64) {
            const __m256i v0 = _mm256_loadu_si256((const __m256i *)p);
            const __m256i v1 = _mm256_loadu_si256((const __m256i *)(p + 32));
            const __m256i f0 = _mm256_or_si256(v0, fold), f1 = _mm256_or_si256(v1, fold);
#define TAOAMD_BITS(lo, hi) ((uint64_t)(uint32_t)_mm256_movemask_epi8(lo) | \
                             ((uint64_t)(uint32_t)_mm256_movemask_epi8(hi) << 32))
            const uint64_t q = TAOAMD_BITS(_mm256_cmpeq_epi8(v0, quote), _mm256_cmpeq_epi8(v1, quote));
            // '{' 0x7b / '[' 0x5b and '}' 0x7d / ']' 0x5d differ in bit 5 only
            const uint64_t o = TAOAMD_BITS(_mm256_cmpeq_epi8(f0, opn), _mm256_cmpeq_epi8(f1, opn));
            const uint64_t c = TAOAMD_BITS(_mm256_cmpeq_epi8(f0, cls), _mm256_cmpeq_epi8(f1, cls));
#undef TAOAMD_BITS
            uint64_t x = q;             // bit i: parity of the quotes at bytes 0 .. i
            x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16; x ^= x << 32;
            const uint64_t in = x ^ inside;
            inside = (in >> 63) ? ~0ull : 0ull;
            for (uint64_t st = (o | c) & ~in; st != 0; st &= st - 1) {
                const int i = __builtin_ctzll(st);
                if ((o >> i) & 1u) open_at(p + i); else close_at(p + i);
            }
        }
        bytes(p, e, inside != 0);
    }
    // `plain`: the caller knows that the chunk holds no backslash
    void run(const char *b, const char *e, bool s, bool plain)
    {
        static const bool wide = __builtin_cpu_supports("avx2");
        d = m = 0;
        segs.push_back(Seg{0, 0, 0, nullptr});
        if (plain && wide) blocks(b, e, s); else bytes(b, e, s);
        delta = d;
        lowest = m;
    }
    // index of the segment of `level`, -1 if the chunk never was that low
    int find(int64_t level) const
    {
        for (size_t k = 0; k < segs.size(); k++)
            if (segs[k].level == level) return (int)k;
        return -1;
    }
};

// Byte ranges of the container elements of a JSON list, `p0` = first byte
// after the list's '['.  Returns the position of the list's closing bracket
// (nullptr: unterminated); *closed_by is that character (']' for a well
// formed list).  Elements that are not containers (numbers, strings) are not
// reported, like the one-pass scan this replaces.
const char *find_elements(const char *p0, const char *e, Ranges &out, char *closed_by)
{
    out.clear();
    *closed_by = 0;
    const size_t len = (size_t)(e - p0);
    int T = omp_get_max_threads();
    if (len < (1u << 20) || T < 2) T = 1;
    if ((size_t)T > len / 65536 + 1) T = (int)(len / 65536 + 1);
    // Chunks of at most 8 MB, walked T at a time; the rounds stop with the one
    // in which the list closes, so a table at the head of a large file costs
    // its own bytes and not the file's (the five tables of the annotation file
    // were five scans to the end of the file).
    const size_t piece = std::min<size_t>(std::max<size_t>(len / (size_t)T, 1), (size_t)8 << 20);
    const size_t nC = std::max<size_t>(1, (len + piece - 1) / piece);
    std::vector<const char *> cb(nC + 1);
    for (size_t t = 0; t <= nC; t++) {
        const char *q = t == nC ? e : p0 + piece * t;
        while (t > 0 && t < nC && q < e && q[-1] == '\\') q++;
        cb[t] = q;
    }
    for (size_t t = 1; t <= nC; t++) if (cb[t] < cb[t - 1]) cb[t] = cb[t - 1];
    std::vector<uint8_t> in_str(nC + 1, 0), plain(nC, 0);
    std::vector<size_t> quotes(nC, 0);
    std::vector<ChunkWalk> walk(nC);
    std::vector<int64_t> base(nC + 1, 0);
    int last = -1;                  // chunk in which the depth reaches -1
    size_t acc = 0;                 // unescaped quotes before the round
    const bool timing = len > ((size_t)1 << 28) && getenv("TAOAMD_INGEST_TIMING") != nullptr;
    double t_a = 0, t_b = 0;
    const double t_begin = omp_get_wtime();
    for (size_t r0 = 0; r0 < nC && last < 0; r0 += (size_t)T) {
        const int n = (int)std::min<size_t>((size_t)T, nC - r0);
        const double t0 = omp_get_wtime();
        // (A) unescaped quotes per chunk -> does a chunk start inside a string
        if (nC > 1) {
#pragma omp parallel for schedule(static, 1) if (n > 1)
            for (int u = 0; u < n; u++) {
                size_t q = 0;
                const char *b = cb[r0 + u], *en = cb[r0 + u + 1];
                if (!memchr(b, '\\', (size_t)(en - b))) {
                    plain[r0 + u] = 1;
                    for (const char *p = b; p < en; p++) q += *p == '"';
                } else {
                    for (const char *p = b; p < en; p++) {
                        if (*p == '\\') p++;
                        else if (*p == '"') q++;
                    }
                }
                quotes[r0 + u] = q;
            }
            for (int u = 0; u < n; u++) { in_str[r0 + u] = acc & 1; acc += quotes[r0 + u]; }
        }
        const double t1 = omp_get_wtime();
        // (B) candidate boundaries of every chunk
#pragma omp parallel for schedule(static, 1) if (n > 1)
        for (int u = 0; u < n; u++)
            walk[r0 + u].run(cb[r0 + u], cb[r0 + u + 1], in_str[r0 + u], plain[r0 + u]);
        t_a += t1 - t0;
        t_b += omp_get_wtime() - t1;
        for (size_t t = r0; t < r0 + (size_t)n; t++) {
            if (base[t] + walk[t].lowest < 0) { last = (int)t; break; }
            base[t + 1] = base[t] + walk[t].delta;
        }
    }
    if (last < 0) return nullptr;
    const int closing = walk[last].find(-base[last] - 1);
    if (closing < 0) return nullptr;
    const char *close_pos = walk[last].segs[closing].opened_by;
    // the true segment of every chunk: [s_lo, s_hi) of its starts, [e_lo, e_hi) of its ends
    std::vector<size_t> s_lo((size_t)last + 1, 0), s_hi((size_t)last + 1, 0),
        e_lo((size_t)last + 1, 0), e_hi((size_t)last + 1, 0), so((size_t)last + 2, 0),
        eo((size_t)last + 2, 0);
    for (int t = 0; t <= last; t++) {
        const ChunkWalk &w = walk[t];
        const int k = w.find(-base[t]);
        if (k >= 0) {
            const bool more = (size_t)k + 1 < w.segs.size();
            s_lo[t] = w.segs[k].s0; s_hi[t] = more ? w.segs[k + 1].s0 : w.starts.size();
            e_lo[t] = w.segs[k].e0; e_hi[t] = more ? w.segs[k + 1].e0 : w.ends.size();
        }
        so[t + 1] = so[t] + (s_hi[t] - s_lo[t]);
        eo[t + 1] = eo[t] + (e_hi[t] - e_lo[t]);
    }
    // (an element may start in one chunk and end in a later one: starts and
    // ends are laid out independently, each chunk's at its own offset)
    if (so[last + 1] != eo[last + 1]) return nullptr;
    out.resize(so[last + 1]);
#pragma omp parallel for schedule(static, 1) if (T > 1)
    for (int t = 0; t <= last; t++) {
        size_t k = so[t];
        for (size_t a = s_lo[t]; a < s_hi[t]; a++) out[k++].first = walk[t].starts[a];
        k = eo[t];
        for (size_t a = e_lo[t]; a < e_hi[t]; a++) out[k++].second = walk[t].ends[a];
    }
    if (timing)
        fprintf(stderr, "taoamd ingest: list of %.2f GB: quotes %.3f s, brackets %.3f s, "
                "collect %.3f s (%zu pieces, %d threads)\n", len / 1e9, t_a, t_b,
                omp_get_wtime() - t_begin - t_a - t_b, nC, T);
    *closed_by = *close_pos;
    return close_pos;
}

// text of a JSON string body with its escapes resolved (ASCII \uXXXX only:
// enough for key names)
static std::string unescape(const char *b, const char *e)
{
    std::string out;
    for (const char *p = b; p < e; p++) {
        if (*p != '\\' || p + 1 >= e) { out.push_back(*p); continue; }
        p++;
        switch (*p) {
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'n': out.push_back('\n'); break;
        case 'r': out.push_back('\r'); break;
        case 't': out.push_back('\t'); break;
        case 'u':
            if (p + 4 < e) {
                unsigned v = 0;
                for (int k = 1; k <= 4; k++) {
                    const char c = p[k];
                    v = v * 16 + (c >= '0' && c <= '9' ? c - '0'
                                  : c >= 'a' && c <= 'f' ? c - 'a' + 10
                                  : c >= 'A' && c <= 'F' ? c - 'A' + 10 : 0);
                }
                out.push_back(v < 128 ? (char)v : '?');
                p += 4;
            }
            break;
        default: out.push_back(*p);
        }
    }
    return out;
}

bool key_is(const char *b, const char *e, const char *name)
{
    size_t n = strlen(name);
    if ((size_t)(e - b) == n && !memcmp(b, name, n)) return true;
    if (!memchr(b, '\\', (size_t)(e - b))) return false;
    return unescape(b, e) == name;        // "\u0069mage_id" is image_id too
}

// the columns as raw arrays (a Columns' vectors, or the caller's numpy arrays)
struct ColView {
    int64_t *image_id, *category_id, *track_id, *video_id;
    double *bbox, *score;
};

// one prediction object [b, e) -> row i of the columns
bool parse_object(const char *b, const char *e, int64_t i, const ColView &c, std::string &err)
{
    Cursor cur{b, e};
    if (!cur.eat('{')) { err = "expected object"; return false; }
    bool has_img = false, has_cat = false, has_box = false, has_score = false;
    c.track_id[i] = -1;
    c.video_id[i] = -1;
    if (!cur.eat('}')) {
        for (;;) {
            const char *kb, *ke;
            if (!cur.str(kb, ke) || !cur.eat(':')) { cur.bad("expected key"); break; }
            double v;
            int64_t iv;
            // the six known keys by length and text; anything else (or a key
            // spelt with escapes) through the general comparison
            const size_t kl = (size_t)(ke - kb);
            int which = -1;
            if (kl == 8) {
                which = !memcmp(kb, "image_id", 8) ? 0 : !memcmp(kb, "track_id", 8) ? 2
                        : !memcmp(kb, "video_id", 8) ? 3 : -1;
            } else if (kl == 11) {
                which = !memcmp(kb, "category_id", 11) ? 1 : -1;
            } else if (kl == 5) {
                which = !memcmp(kb, "score", 5) ? 4 : -1;
            } else if (kl == 4) {
                which = !memcmp(kb, "bbox", 4) ? 5 : -1;
            }
            if (which < 0 && memchr(kb, '\\', kl)) {
                static const char *const names[6] = {"image_id", "category_id", "track_id",
                                                     "video_id", "score", "bbox"};
                for (int w = 0; w < 6 && which < 0; w++)
                    if (key_is(kb, ke, names[w])) which = w;
            }
            if (which == 0) { if (cur.integer(iv)) { c.image_id[i] = iv; has_img = true; } }
            else if (which == 1) { if (cur.integer(iv)) { c.category_id[i] = iv; has_cat = true; } }
            else if (which == 2) { if (cur.integer(iv)) c.track_id[i] = iv; }
            else if (which == 3) { if (cur.integer(iv)) c.video_id[i] = iv; }
            else if (which == 4) {
                cur.ws();
                if (cur.p < cur.e && (*cur.p == 't' || *cur.p == 'f')) {
                    c.score[i] = *cur.p == 't' ? 1.0 : 0.0;     // True == 1
                    cur.skip();
                    has_score = true;
                } else if (cur.number(v)) { c.score[i] = v; has_score = true; }
            }
            else if (which == 5) {
                if (!cur.eat('[')) cur.bad("bbox is not a list");
                for (int k = 0; k < 4 && !cur.fail; k++) {
                    if (k && !cur.eat(',')) cur.bad("bbox needs 4 numbers");
                    if (cur.number(v)) c.bbox[4 * i + k] = v;
                }
                if (!cur.fail && !cur.eat(']')) cur.bad("bbox needs 4 numbers");
                has_box = !cur.fail;
            } else cur.skip();
            if (cur.fail) break;
            if (cur.eat(',')) continue;
            if (cur.eat('}')) break;
            cur.bad("expected , or }");
            break;
        }
    }
    if (cur.fail) { err = cur.why; return false; }
    if (!has_img) { err = "KeyError: 'image_id'"; return false; }
    if (!has_cat) { err = "KeyError: 'category_id'"; return false; }
    if (!has_box) { err = "KeyError: 'bbox'"; return false; }
    if (!has_score) { err = "KeyError: 'score'"; return false; }
    return true;
}


// ---------------------------------------------------------------------------
// annotation file
// ---------------------------------------------------------------------------
struct GT {
    std::vector<int64_t> cat_id, cat_merged;
    std::vector<uint8_t> cat_freq;
    std::vector<int64_t> vid_id, vid_neg_off, vid_neg, vid_nel_off, vid_nel;
    // (per-image / per-annotation columns: filled by parallel loops, Vec)
    Vec<int64_t> img_id, img_vid;
    std::vector<int64_t> img_neg_off, img_neg, img_nel_off, img_nel;
    Vec<double> img_frame;
    std::vector<int64_t> trk_id, trk_cat, trk_vid;
    std::vector<uint8_t> trk_ignore;
    Vec<int64_t> ann_id, ann_img, ann_trk, ann_cat;
    Vec<double> ann_bbox, ann_area, ann_vis;
    std::vector<uint8_t> ann_oof, ann_ignore;
};


// Python truthiness of a JSON value (`1 if a.get("ignore", 0) else 0`)
bool truthy(Cursor &c)
{
    c.ws();
    if (c.p >= c.e) { c.bad("unexpected end"); return false; }
    const char ch = *c.p;
    if (ch == 't') { c.skip(); return true; }
    if (ch == 'f' || ch == 'n') { c.skip(); return false; }
    if (ch == '"') { const char *b = nullptr, *e = nullptr; c.str(b, e); return e > b; }
    if (ch == '[' || ch == '{') {
        const char *q = c.p + 1;
        while (q < c.e && (*q == ' ' || *q == '\n' || *q == '\t' || *q == '\r')) q++;
        const bool empty = q < c.e && (*q == ']' || *q == '}');
        c.skip();
        return !empty;
    }
    double v;
    if (!c.number(v)) return false;
    return v != 0.0;   // NaN is truthy, as in Python
}

// [ obj, obj, ... ] -> byte range of every element; the cursor ends after ']'
bool scan_array(Cursor &c, Ranges &out)
{
    out.clear();
    if (!c.eat('[')) { c.bad("table is not a list"); return false; }
    char closer = 0;
    const char *close_pos = find_elements(c.p, c.e, out, &closer);
    if (!close_pos || closer != ']') { c.bad("unterminated list"); return false; }
    c.p = close_pos + 1;
    return true;
}

bool id_list(Cursor &c, std::vector<int64_t> &out)
{
    out.clear();
    if (!c.eat('[')) { c.bad("expected a list of ids"); return false; }
    if (c.eat(']')) return true;
    for (;;) {
        int64_t v;
        if (!c.integer(v)) return false;
        out.push_back(v);
        if (c.eat(',')) continue;
        if (c.eat(']')) return true;
        c.bad("expected , or ]");
        return false;
    }
}

// generic object walk: f(key_begin, key_end, cursor) consumes the value
template <class F>
bool walk_object(const char *b, const char *e, std::string &err, F f)
{
    Cursor cur{b, e};
    if (!cur.eat('{')) { err = "table element is not an object"; return false; }
    if (!cur.eat('}')) {
        for (;;) {
            const char *kb, *ke;
            if (!cur.str(kb, ke) || !cur.eat(':')) { cur.bad("expected key"); break; }
            f(kb, ke, cur);
            if (cur.fail) break;
            if (cur.eat(',')) continue;
            if (cur.eat('}')) break;
            cur.bad("expected , or }");
            break;
        }
    }
    if (cur.fail) { err = cur.why; return false; }
    return true;
}

void csr(const std::vector<std::vector<int64_t>> &lists, std::vector<int64_t> &off,
         std::vector<int64_t> &val)
{
    off.assign(lists.size() + 1, 0);
    for (size_t i = 0; i < lists.size(); i++) off[i + 1] = off[i] + (int64_t)lists[i].size();
    val.resize((size_t)off.back());
    for (size_t i = 0; i < lists.size(); i++)
        std::copy(lists[i].begin(), lists[i].end(), val.begin() + off[i]);
}

struct Fail {
    bool ok = true;
    std::string msg;
    void set(const std::string &table, int64_t i, const std::string &m)
    {
#pragma omp critical(taoamd_gt_fail)
        { if (ok) { ok = false; msg = m.rfind("KeyError", 0) == 0 ? m : table + " " + std::to_string(i) + ": " + m; } }
    }
};

#define NEED(flag, name) if (!(flag)) { fl.set(table, i, "KeyError: '" name "'"); continue; }

void parse_categories(const Ranges &r, GT &g, Fail &fl)
{
    const char *table = "category";
    const int64_t n = (int64_t)r.size();
    g.cat_id.resize(n); g.cat_freq.assign(n, (uint8_t)'?');
    std::vector<std::vector<int64_t>> merged(n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        bool has_id = false;
        std::string er;
        const bool ok = walk_object(r[i].first, r[i].second, er, [&](const char *kb, const char *ke, Cursor &c) {
            int64_t v;
            if (key_is(kb, ke, "id")) { if (c.integer(v)) { g.cat_id[i] = v; has_id = true; } }
            else if (key_is(kb, ke, "frequency")) {
                const char *b, *e;
                if (c.str(b, e)) {
                    // the letter; 0xFF for any other text ("rare", ""): the
                    // reference's img_count_lbl.index() fails on those
                    const bool letter = e - b == 1 && *b != '\\' && *b != '?' &&
                                        (unsigned char)*b < 0x80;
                    g.cat_freq[i] = letter ? (uint8_t)*b : (uint8_t)0xFF;
                }
            } else if (key_is(kb, ke, "merged")) {
                Ranges ms;
                merged[i].clear();
                if (scan_array(c, ms))
                    for (auto &m : ms) {
                        bool got = false;
                        std::string e2;
                        if (!walk_object(m.first, m.second, e2, [&](const char *b2, const char *e3, Cursor &c2) {
                                int64_t w;
                                if (key_is(b2, e3, "id")) { if (c2.integer(w)) { merged[i].push_back(w); got = true; } }
                                else c2.skip();
                            })) { c.bad("malformed merged entry"); break; }
                        if (!got) { c.bad("KeyError: 'id'"); break; }
                    }
            } else c.skip();
        });
        if (!ok) { fl.set(table, i, er); continue; }
        NEED(has_id, "id");
    }
    for (int64_t i = 0; i < n; i++)
        for (int64_t m : merged[i]) { g.cat_merged.push_back(m); g.cat_merged.push_back(g.cat_id[i]); }
}

void parse_videos(const Ranges &r, GT &g, Fail &fl)
{
    const char *table = "video";
    const int64_t n = (int64_t)r.size();
    g.vid_id.resize(n);
    std::vector<std::vector<int64_t>> neg(n), nel(n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        bool has_id = false, has_neg = false, has_nel = false;
        std::string er;
        const bool ok = walk_object(r[i].first, r[i].second, er, [&](const char *kb, const char *ke, Cursor &c) {
            int64_t v;
            if (key_is(kb, ke, "id")) { if (c.integer(v)) { g.vid_id[i] = v; has_id = true; } }
            else if (key_is(kb, ke, "neg_category_ids")) has_neg = id_list(c, neg[i]);
            else if (key_is(kb, ke, "not_exhaustive_category_ids")) has_nel = id_list(c, nel[i]);
            else c.skip();
        });
        if (!ok) { fl.set(table, i, er); continue; }
        NEED(has_id, "id"); NEED(has_neg, "neg_category_ids");
        NEED(has_nel, "not_exhaustive_category_ids");
    }
    csr(neg, g.vid_neg_off, g.vid_neg);
    csr(nel, g.vid_nel_off, g.vid_nel);
}

void parse_images(const Ranges &r, GT &g, Fail &fl)
{
    const char *table = "image";
    const int64_t n = (int64_t)r.size();
    g.img_id.resize(n); g.img_vid.resize(n); g.img_frame.resize(n);
    std::vector<std::vector<int64_t>> neg(n), nel(n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        bool has_id = false, has_vid = false, has_fr = false, has_neg = false, has_nel = false;
        std::string er;
        const bool ok = walk_object(r[i].first, r[i].second, er, [&](const char *kb, const char *ke, Cursor &c) {
            int64_t v;
            double d;
            if (key_is(kb, ke, "id")) { if (c.integer(v)) { g.img_id[i] = v; has_id = true; } }
            else if (key_is(kb, ke, "video_id")) { if (c.integer(v)) { g.img_vid[i] = v; has_vid = true; } }
            else if (key_is(kb, ke, "frame_index")) { if (c.number(d)) { g.img_frame[i] = d; has_fr = true; } }
            else if (key_is(kb, ke, "neg_category_ids")) has_neg = id_list(c, neg[i]);
            else if (key_is(kb, ke, "not_exhaustive_category_ids")) has_nel = id_list(c, nel[i]);
            else c.skip();
        });
        if (!ok) { fl.set(table, i, er); continue; }
        NEED(has_id, "id"); NEED(has_vid, "video_id"); NEED(has_fr, "frame_index");
        NEED(has_neg, "neg_category_ids"); NEED(has_nel, "not_exhaustive_category_ids");
    }
    csr(neg, g.img_neg_off, g.img_neg);
    csr(nel, g.img_nel_off, g.img_nel);
}

void parse_tracks(const Ranges &r, GT &g, Fail &fl)
{
    const char *table = "track";
    const int64_t n = (int64_t)r.size();
    g.trk_id.resize(n); g.trk_cat.resize(n); g.trk_vid.resize(n); g.trk_ignore.assign(n, 0);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        bool has_id = false, has_cat = false, has_vid = false;
        std::string er;
        const bool ok = walk_object(r[i].first, r[i].second, er, [&](const char *kb, const char *ke, Cursor &c) {
            int64_t v;
            if (key_is(kb, ke, "id")) { if (c.integer(v)) { g.trk_id[i] = v; has_id = true; } }
            else if (key_is(kb, ke, "category_id")) { if (c.integer(v)) { g.trk_cat[i] = v; has_cat = true; } }
            else if (key_is(kb, ke, "video_id")) { if (c.integer(v)) { g.trk_vid[i] = v; has_vid = true; } }
            else if (key_is(kb, ke, "ignore")) g.trk_ignore[i] = truthy(c) ? 1 : 0;
            else c.skip();
        });
        if (!ok) { fl.set(table, i, er); continue; }
        NEED(has_id, "id"); NEED(has_cat, "category_id"); NEED(has_vid, "video_id");
    }
}

void parse_annotations(const Ranges &r, GT &g, Fail &fl)
{
    const char *table = "annotation";
    const int64_t n = (int64_t)r.size();
    g.ann_id.resize(n); g.ann_img.resize(n); g.ann_trk.resize(n); g.ann_cat.resize(n);
    g.ann_bbox.resize(4 * n); g.ann_area.resize(n); g.ann_vis.resize(n);
    g.ann_oof.assign(n, 0); g.ann_ignore.assign(n, 0);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        bool has_id = false, has_img = false, has_trk = false, has_cat = false,
             has_box = false, has_area = false, has_vis = false, has_oof = false;
        std::string er;
        const bool ok = walk_object(r[i].first, r[i].second, er, [&](const char *kb, const char *ke, Cursor &c) {
            int64_t v;
            double d;
            if (key_is(kb, ke, "id")) { if (c.integer(v)) { g.ann_id[i] = v; has_id = true; } }
            else if (key_is(kb, ke, "image_id")) { if (c.integer(v)) { g.ann_img[i] = v; has_img = true; } }
            else if (key_is(kb, ke, "track_id")) { if (c.integer(v)) { g.ann_trk[i] = v; has_trk = true; } }
            else if (key_is(kb, ke, "category_id")) { if (c.integer(v)) { g.ann_cat[i] = v; has_cat = true; } }
            else if (key_is(kb, ke, "area")) { if (c.number(d)) { g.ann_area[i] = d; has_area = true; } }
            else if (key_is(kb, ke, "visibility")) { if (c.number(d)) { g.ann_vis[i] = d; has_vis = true; } }
            else if (key_is(kb, ke, "out_of_frame")) { g.ann_oof[i] = truthy(c) ? 1 : 0; has_oof = true; }
            else if (key_is(kb, ke, "ignore")) g.ann_ignore[i] = truthy(c) ? 1 : 0;
            else if (key_is(kb, ke, "bbox")) {
                if (!c.eat('[')) c.bad("bbox is not a list");
                for (int k = 0; k < 4 && !c.fail; k++) {
                    if (k && !c.eat(',')) c.bad("bbox needs 4 numbers");
                    if (c.number(d)) g.ann_bbox[4 * i + k] = d;
                }
                if (!c.fail && !c.eat(']')) c.bad("bbox needs 4 numbers");
                has_box = !c.fail;
            } else c.skip();
        });
        if (!ok) { fl.set(table, i, er); continue; }
        NEED(has_id, "id"); NEED(has_img, "image_id"); NEED(has_trk, "track_id");
        NEED(has_cat, "category_id"); NEED(has_box, "bbox"); NEED(has_area, "area");
        NEED(has_vis, "visibility"); NEED(has_oof, "out_of_frame");
    }
}

}  // namespace

extern "C" {

// A prediction file scanned: the mapping, the byte range of every element and
// the share [i0, i1) of them this process converts.
struct PredScan {
    const char *buf = nullptr;
    size_t len = 0;
    Ranges objs;
    int64_t i0 = 0, i1 = 0;
    ~PredScan() { if (len) munmap((void *)buf, len); }
};

static PredScan *pred_scan(const char *path, int64_t part, int64_t n_parts, char *err,
                           size_t errlen)
{
    auto fail = [&](const std::string &m) -> PredScan * {
        if (err && errlen) snprintf(err, errlen, "%s", m.c_str());
        return nullptr;
    };
    if (n_parts < 1 || part < 0 || part >= n_parts) return fail("bad part");
    const bool timing = getenv("TAOAMD_INGEST_TIMING") != nullptr;
    const double t0 = omp_get_wtime();
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(std::string("cannot open ") + path);
    struct stat st;
    fstat(fd, &st);
    std::unique_ptr<PredScan> ps(new PredScan);
    const size_t len = (size_t)st.st_size;
    const char *buf = len ? (const char *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0) : "";
    close(fd);
    if (len && buf == MAP_FAILED) return fail("mmap failed");
    ps->buf = buf;
    ps->len = len;
    const char *p = buf, *e = buf + len;
    while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++;
    if (p >= e || *p != '[') return fail("results is not a list.");
    p++;
    // byte range of every element of the list
    Ranges &el = ps->objs;
    char closer = 0;
    const char *close_pos = find_elements(p, e, el, &closer);
    if (!close_pos || closer != ']') return fail("unterminated list");
    const double t1 = omp_get_wtime();
    // between the objects only commas and white space may stand: a bare
    // number or string in the list is not a prediction (json.load would hand
    // it to the evaluator, which fails on it), and nothing but white space may
    // follow the list
    auto blank = [](const char *a, const char *b, bool commas) {
        for (; a < b; a++)
            if (!(*a == ' ' || *a == '\n' || *a == '\t' || *a == '\r' || (commas && *a == ',')))
                return false;
        return true;
    };
    if (!blank(close_pos + 1, e, false)) return fail("Extra data after the list");
    const int64_t n = (int64_t)el.size();
    bool clean = true;
#pragma omp parallel for schedule(static) reduction(&& : clean)
    for (int64_t i = 0; i < n; i++)
        clean = clean && blank(i ? el[i - 1].second : p, el[i].first, true);
    if (clean) clean = blank(n ? el[n - 1].second : p, close_pos, true);
    if (!clean) return fail("list element is not an object");
    ps->i0 = n * part / n_parts;
    ps->i1 = n * (part + 1) / n_parts;
    if (timing)
        fprintf(stderr, "taoamd ingest: %s: elements %.3f s, gaps %.3f s (%d threads)\n", path,
                t1 - t0, omp_get_wtime() - t1, omp_get_max_threads());
    return ps.release();
}

// the share's numbers into the arrays of `c` (every thread touches its own
// rows first: freshly allocated arrays are faulted in by all cores)
static bool pred_convert(const PredScan &ps, const ColView &c, std::string &first_err)
{
    bool ok = true;
    const int64_t n = ps.i1 - ps.i0;
    const double t0 = omp_get_wtime();
    // the caller's freshly allocated columns (2.2 GB at 30 M predictions) are
    // touched here for the first time: huge pages where the system hands them out
    // on request (transparent_hugepage = madvise on the MI355X boxes: first
    // touch of 2.2 GB 0.13 s with 4 KB pages, 0.01 s with 2 MB ones)
    auto huge = [](void *p, size_t bytes) {
        const uintptr_t a = ((uintptr_t)p + ((size_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1);
        const uintptr_t b = ((uintptr_t)p + bytes) & ~(((uintptr_t)2 << 20) - 1);
        if (b > a) madvise((void *)a, b - a, MADV_HUGEPAGE);
    };
    for (int64_t *col : {c.image_id, c.category_id, c.track_id, c.video_id})
        huge(col, (size_t)n * 8);
    huge(c.bbox, (size_t)n * 32);
    huge(c.score, (size_t)n * 8);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        std::string er;
        const Span &o = ps.objs[ps.i0 + i];
        if (*o.first != '{') er = "list element is not an object";
        if (!er.empty() || !parse_object(o.first, o.second, i, c, er)) {
#pragma omp critical
            { if (ok) { ok = false; first_err = "prediction " + std::to_string(ps.i0 + i) + ": " + er; } }
        }
    }
    if (getenv("TAOAMD_INGEST_TIMING"))
        fprintf(stderr, "taoamd ingest: convert %lld predictions %.3f s\n", (long long)n,
                omp_get_wtime() - t0);
    return ok;
}

static void *pred_parse_impl(const char *path, int64_t part, int64_t n_parts, char *err,
                             size_t errlen)
{
    const bool timing = getenv("TAOAMD_INGEST_TIMING") != nullptr;
    const double t_begin = omp_get_wtime();
    std::unique_ptr<PredScan> ps(pred_scan(path, part, n_parts, err, errlen));
    if (!ps) return nullptr;
    const double t_scan = omp_get_wtime();
    std::unique_ptr<Columns> c(new Columns);
    const int64_t n = ps->i1 - ps->i0;
    c->first = ps->i0;
    c->total = (int64_t)ps->objs.size();
    c->image_id.resize(n); c->category_id.resize(n); c->track_id.resize(n);
    c->video_id.resize(n); c->score.resize(n); c->bbox.resize(4 * n);
    const ColView v{c->image_id.data(), c->category_id.data(), c->track_id.data(),
                    c->video_id.data(), c->bbox.data(), c->score.data()};
    std::string first_err;
    const bool ok = pred_convert(*ps, v, first_err);
    if (timing)
        fprintf(stderr, "taoamd ingest: %s: scan %.3f s, convert %.3f s (%d threads)\n", path,
                t_scan - t_begin, omp_get_wtime() - t_scan, omp_get_max_threads());
    if (!ok) {
        if (err && errlen) snprintf(err, errlen, "%s", first_err.c_str());
        return nullptr;
    }
    return c.release();
}

void *taoamd_pred_parse(const char *path, char *err, size_t errlen)
{
    taoamd::ThreadScope threads;
    return pred_parse_impl(path, 0, 1, err, errlen);
}

// One share of the list for one rank of a multi-process run: the structural
// scan covers the whole file (exact element boundaries, OpenMP), the numbers of
// elements [n * part / n_parts, n * (part + 1) / n_parts) only are converted.
void *taoamd_pred_parse_part(const char *path, int64_t part, int64_t n_parts, char *err,
                             size_t errlen)
{
    taoamd::ThreadScope threads;
    return pred_parse_impl(path, part, n_parts, err, errlen);
}

// The same in two steps, the numbers converted straight into the caller's
// arrays (no intermediate copy: 56 bytes a prediction): scan, ask for the
// share's length, convert, free.
void *taoamd_pred_scan(const char *path, int64_t part, int64_t n_parts, char *err, size_t errlen)
{
    taoamd::ThreadScope threads;
    return pred_scan(path, part, n_parts, err, errlen);
}

void taoamd_pred_scan_info(void *h, int64_t *first, int64_t *count, int64_t *total)
{
    const PredScan *ps = (const PredScan *)h;
    if (first) *first = ps->i0;
    if (count) *count = ps->i1 - ps->i0;
    if (total) *total = (int64_t)ps->objs.size();
}

int taoamd_pred_convert(void *h, int64_t *image_id, int64_t *category_id, double *bbox,
                        double *score, int64_t *track_id, int64_t *video_id, char *err,
                        size_t errlen)
{
    taoamd::ThreadScope threads;
    const PredScan *ps = (const PredScan *)h;
    if (ps->i1 > ps->i0 && (!image_id || !category_id || !bbox || !score || !track_id || !video_id))
        return 1;
    const ColView v{image_id, category_id, track_id, video_id, bbox, score};
    std::string first_err;
    if (!pred_convert(*ps, v, first_err)) {
        if (err && errlen) snprintf(err, errlen, "%s", first_err.c_str());
        return 2;
    }
    return 0;
}

void taoamd_pred_scan_free(void *h) { delete (PredScan *)h; }

// Rows the device-side reader (csrc/json_ingest.hip) left to this one: object
// idx[k] of the list starts at byte at[k] of the file; it is read by
// parse_object -- the text of its value, its errors -- into row idx[k] of the
// columns.  0 = ok, 2 = malformed record (the FIRST such object's message in
// err, as taoamd_pred_convert reports it).
int taoamd_pred_patch(const char *path, int64_t n, const int64_t *idx, const int64_t *at,
                      int64_t *image_id, int64_t *category_id, double *bbox, double *score,
                      int64_t *track_id, int64_t *video_id, char *err, size_t errlen)
{
    if (n <= 0) return 0;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return 1;
    struct stat st;
    fstat(fd, &st);
    const size_t len = (size_t)st.st_size;
    const char *buf = (const char *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (buf == MAP_FAILED) return 1;
    const ColView v{image_id, category_id, track_id, video_id, bbox, score};
    int64_t bad = -1;
    std::string bad_msg;
    for (int64_t k = 0; k < n; k++) {
        std::string er;
        bool ok = at[k] >= 0 && (size_t)at[k] < len && buf[at[k]] == '{';
        if (ok) {
            Cursor cur{buf + at[k], buf + len};
            cur.skip();                               // the object's extent
            ok = !cur.fail && parse_object(buf + at[k], cur.p, idx[k], v, er);
            if (cur.fail) er = cur.why;
        } else er = "list element is not an object";
        if (!ok && (bad < 0 || idx[k] < bad)) {
            bad = idx[k];
            bad_msg = "prediction " + std::to_string(idx[k]) + ": " + er;
        }
    }
    munmap((void *)buf, len);
    if (bad >= 0) {
        if (err && errlen) snprintf(err, errlen, "%s", bad_msg.c_str());
        return 2;
    }
    return 0;
}

int64_t taoamd_pred_count(void *h) { return (int64_t)((Columns *)h)->image_id.size(); }

// position of the share's first element in the whole list, and the list's length
void taoamd_pred_part_info(void *h, int64_t *first, int64_t *total)
{
    if (first) *first = ((Columns *)h)->first;
    if (total) *total = ((Columns *)h)->total;
}

int taoamd_pred_copy(void *h, int64_t *image_id, int64_t *category_id, double *bbox,
                     double *score, int64_t *track_id, int64_t *video_id)
{
    taoamd::ThreadScope threads;
    Columns *c = (Columns *)h;
    size_t n = c->image_id.size();
    memcpy(image_id, c->image_id.data(), n * 8);
    memcpy(category_id, c->category_id.data(), n * 8);
    memcpy(bbox, c->bbox.data(), n * 32);
    memcpy(score, c->score.data(), n * 8);
    memcpy(track_id, c->track_id.data(), n * 8);
    memcpy(video_id, c->video_id.data(), n * 8);
    return 0;
}

void taoamd_pred_free(void *h) { delete (Columns *)h; }

// Annotation file -> handle of GTColumns arrays (NULL + message on error;
// "KeyError: 'x'" when a required key is absent, as the reference's dict
// accesses would raise).
void *taoamd_gt_parse(const char *path, char *err, size_t errlen)
{
    taoamd::ThreadScope threads;
    auto fail = [&](const std::string &m) -> void * {
        if (err && errlen) snprintf(err, errlen, "%s", m.c_str());
        return nullptr;
    };
    const bool timing = getenv("TAOAMD_INGEST_TIMING") != nullptr;
    double tt[8];
    tt[0] = omp_get_wtime();
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(std::string("cannot open ") + path);
    struct stat st;
    fstat(fd, &st);
    size_t len = (size_t)st.st_size;
    const char *buf = len ? (const char *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0) : "";
    close(fd);
    if (len && buf == MAP_FAILED) return fail("mmap failed");
    auto done = [&]() { if (len) munmap((void *)buf, len); };
    Cursor cur{buf, buf + len};
    cur.ws();
    if (cur.p >= cur.e || *cur.p != '{') {
        const bool is_list = cur.p < cur.e && *cur.p == '[';
        done();
        return fail(is_list ? "not a dict: list" : "not a dict");
    }
    cur.p++;
    Ranges cats, vids, imgs, trks, anns;
    bool h_cats = false, h_vids = false, h_imgs = false, h_trks = false, h_anns = false;
    if (!cur.eat('}')) {
        for (;;) {
            const char *kb, *ke;
            if (!cur.str(kb, ke) || !cur.eat(':')) { cur.bad("expected key"); break; }
            if (key_is(kb, ke, "categories")) h_cats = scan_array(cur, cats);
            else if (key_is(kb, ke, "videos")) h_vids = scan_array(cur, vids);
            else if (key_is(kb, ke, "images")) h_imgs = scan_array(cur, imgs);
            else if (key_is(kb, ke, "tracks")) h_trks = scan_array(cur, trks);
            else if (key_is(kb, ke, "annotations")) h_anns = scan_array(cur, anns);
            else cur.skip();
            if (cur.fail) break;
            if (cur.eat(',')) continue;
            if (cur.eat('}')) break;
            cur.bad("expected , or }");
            break;
        }
    }
    if (cur.fail) { std::string w = cur.why; done(); return fail("malformed annotation file: " + w); }
    const char *missing = !h_cats ? "categories" : !h_vids ? "videos" : !h_imgs ? "images"
                          : !h_trks ? "tracks" : !h_anns ? "annotations" : nullptr;
    if (missing) { done(); return fail(std::string("KeyError: '") + missing + "'"); }
    GT *g = new GT;
    Fail fl;
    tt[1] = omp_get_wtime();
    parse_categories(cats, *g, fl);
    parse_videos(vids, *g, fl);
    tt[2] = omp_get_wtime();
    parse_images(imgs, *g, fl);
    tt[3] = omp_get_wtime();
    parse_tracks(trks, *g, fl);
    tt[4] = omp_get_wtime();
    parse_annotations(anns, *g, fl);
    tt[5] = omp_get_wtime();
    done();
    if (timing)
        fprintf(stderr, "taoamd ingest: %s: scan %.3f s, categories+videos %.3f, images %.3f, "
                "tracks %.3f, annotations %.3f, unmap %.3f\n", path, tt[1] - tt[0], tt[2] - tt[1],
                tt[3] - tt[2], tt[4] - tt[3], tt[5] - tt[4], omp_get_wtime() - tt[5]);
    if (!fl.ok) { delete g; return fail(fl.msg); }
    return g;
}

// one named array of the handle: pointer, element count, element size
int taoamd_gt_array(void *h, const char *name, const void **ptr, int64_t *count, int *elem)
{
    GT *g = (GT *)h;
#define I64(f) if (!strcmp(name, #f)) { *ptr = g->f.data(); *count = (int64_t)g->f.size(); *elem = 8; return 0; }
#define F64(f) if (!strcmp(name, #f)) { *ptr = g->f.data(); *count = (int64_t)g->f.size(); *elem = -8; return 0; }
#define U8(f) if (!strcmp(name, #f)) { *ptr = g->f.data(); *count = (int64_t)g->f.size(); *elem = 1; return 0; }
    I64(cat_id) I64(cat_merged) U8(cat_freq)
    I64(vid_id) I64(vid_neg_off) I64(vid_neg) I64(vid_nel_off) I64(vid_nel)
    I64(img_id) I64(img_vid) F64(img_frame) I64(img_neg_off) I64(img_neg) I64(img_nel_off) I64(img_nel)
    I64(trk_id) I64(trk_cat) I64(trk_vid) U8(trk_ignore)
    I64(ann_id) I64(ann_img) I64(ann_trk) I64(ann_cat) F64(ann_bbox) F64(ann_area) F64(ann_vis)
    U8(ann_oof) U8(ann_ignore)
#undef I64
#undef F64
#undef U8
    return -1;
}

// copy of one named array into caller memory, by all cores (the destination's
// pages are first touched in parallel)
int taoamd_gt_copy(void *h, const char *name, void *dst)
{
    taoamd::ThreadScope threads;
    const void *src;
    int64_t count;
    int elem;
    if (taoamd_gt_array(h, name, &src, &count, &elem)) return -1;
    const int64_t bytes = count * (elem < 0 ? -elem : elem);
    const int64_t chunk = 1 << 20;
#pragma omp parallel for schedule(static)
    for (int64_t o = 0; o < bytes; o += chunk)
        memcpy((char *)dst + o, (const char *)src + o, (size_t)std::min(chunk, bytes - o));
    return 0;
}

void taoamd_gt_free(void *h) { delete (GT *)h; }

int taoamd_host_threads(void) { return taoamd::host_threads(); }

int taoamd_host_thread_cap(int n)
{
    const int before = taoamd::g_thread_cap;
    taoamd::g_thread_cap = n > 0 ? n : 0;
    return before;
}

// 1 if every value occurs in `keys` (ascending), 0 if one does not: the
// "Results do not correspond to current LVIS set." test of LVISResults
// (reference lvis_amodal/results.py:62-65) over 30 M image ids on all threads.
int taoamd_host_all_in_sorted(int64_t n_keys, const int64_t *keys, int64_t n,
                              const int64_t *values)
{
    taoamd::ThreadScope threads;
    if (n_keys < 0 || n < 0 || (n_keys && !keys) || (n && !values)) return -1;
    if (n == 0) return 1;
    if (n_keys == 0) return 0;
    const int64_t lo = keys[0], hi = keys[n_keys - 1];
    int missing = 0;
    // ids in a modest range: membership bits; otherwise a binary search each
    if ((uint64_t)(hi - lo) < (uint64_t)1 << 31) {
        std::vector<uint8_t> has((size_t)(hi - lo) + 1, 0);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n_keys; i++) has[(size_t)(keys[i] - lo)] = 1;
#pragma omp parallel for schedule(static) reduction(| : missing)
        for (int64_t i = 0; i < n; i++) {
            const int64_t v = values[i];
            missing |= (v < lo || v > hi || !has[(size_t)(v - lo)]) ? 1 : 0;
        }
    } else {
#pragma omp parallel for schedule(static) reduction(| : missing)
        for (int64_t i = 0; i < n; i++)
            missing |= std::binary_search(keys, keys + n_keys, values[i]) ? 0 : 1;
    }
    return missing ? 0 : 1;
}

// order[] = np.lexsort((arange(n), -score, key)): ascending key, descending
// score inside a key (NaN scores last, -0.0 == 0.0), input order on ties.
// score may be NULL (plain stable argsort of key).  Records are sorted by
// value (no indirect comparisons) with the parallel stable merge sort of
// libstdc++.
int taoamd_host_sort_key_score(int64_t n, const int64_t *key, const double *score,
                               int64_t *order)
{
    taoamd::ThreadScope threads;
    if (n < 0 || (n > 0 && (!key || !order))) return 1;
    if (!score && n >= 65536 && n < ((int64_t)1 << 31)) {
        // Keys alone (the cell keys of the ground truth: category * units +
        // unit, below 2^31): a parallel LSD radix sort of (key, index) pairs,
        // 8 bits a pass, as many passes as the largest key has bytes -- 3 M
        // keys in ~15 ms where the merge sort of 24-byte records takes ~100
        // (round 4: it was the largest single item of the ground-truth halves
        // of the cell tables, which sit on the CLI's critical path).
        const int T = std::max(1, std::min(32, taoamd::team_threads()));
        int64_t lo = INT64_MAX, hi = INT64_MIN;
#pragma omp parallel for schedule(static) num_threads(T) reduction(min : lo) reduction(max : hi)
        for (int64_t i = 0; i < n; i++) {
            lo = std::min(lo, key[i]);
            hi = std::max(hi, key[i]);
        }
        if (lo >= 0 && hi < ((int64_t)1 << 32)) {
            struct P { uint32_t k, i; };
            std::vector<P, NoInit<P>> a((size_t)n), b((size_t)n);
#pragma omp parallel for schedule(static) num_threads(T)
            for (int64_t i = 0; i < n; i++) a[i] = P{(uint32_t)key[i], (uint32_t)i};
            std::vector<int64_t> hist((size_t)T * 256);
            P *src = a.data(), *dst = b.data();
            for (int shift = 0; shift < 32 && (hi >> shift) != 0; shift += 8) {
#pragma omp parallel num_threads(T)
                {
                    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
                    const int64_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
                    int64_t *h = hist.data() + (size_t)t * 256;
                    std::fill(h, h + 256, 0);
                    for (int64_t i = i0; i < i1; i++) h[(src[i].k >> shift) & 255]++;
#pragma omp barrier
#pragma omp single
                    {
                        int64_t run = 0;       // (digit, thread) order: stable
                        for (int d = 0; d < 256; d++)
                            for (int u = 0; u < nt; u++) {
                                int64_t &c = hist[(size_t)u * 256 + d];
                                const int64_t v = c;
                                c = run;
                                run += v;
                            }
                    }
                    for (int64_t i = i0; i < i1; i++) dst[h[(src[i].k >> shift) & 255]++] = src[i];
                }
                std::swap(src, dst);
            }
#pragma omp parallel for schedule(static) num_threads(T)
            for (int64_t i = 0; i < n; i++) order[i] = src[i].i;
            return 0;
        }
    }
    struct Rec { int64_t key; double neg; int64_t idx; };
    std::vector<Rec> r((size_t)n);
#pragma omp parallel for schedule(static) num_threads(std::min(32, taoamd::team_threads()))
    for (int64_t i = 0; i < n; i++) r[i] = Rec{key[i], score ? -score[i] : 0.0, i};
    // a team sized to the input: on a 256-core host the full team costs more
    // in start-up and merge steps than it saves below a few million records
    int team = (int)std::min<int64_t>(32, std::max<int64_t>(1, n / 65536));
    team = std::min(team, omp_get_max_threads());
    const __gnu_parallel::multiway_mergesort_tag tag(team);
    if (score)
        __gnu_parallel::stable_sort(r.begin(), r.end(), [](const Rec &a, const Rec &b) {
            if (a.key != b.key) return a.key < b.key;
            return a.neg < b.neg || (b.neg != b.neg && a.neg == a.neg);
        }, tag);
    else
        __gnu_parallel::stable_sort(r.begin(), r.end(),
                                    [](const Rec &a, const Rec &b) { return a.key < b.key; },
                                    tag);
#pragma omp parallel for schedule(static) num_threads(std::min(32, taoamd::team_threads()))
    for (int64_t i = 0; i < n; i++) order[i] = r[i].idx;
    return 0;
}

// out[i] = index of values[i] in `keys` (ascending, unique), -1 when absent --
// flatten._lookup on all threads (ids in a modest range through a dense table,
// else a binary search each).  The ground-truth halves of the cell tables look
// up 3 M image / category / track ids a dozen times.
int taoamd_host_lookup(int64_t n_keys, const int64_t *keys, int64_t n,
                       const int64_t *values, int64_t *out)
{
    taoamd::ThreadScope threads;
    if (n_keys < 0 || n < 0 || (n_keys && !keys) || (n && (!values || !out))) return 1;
    const int T = std::max(1, std::min(32, taoamd::team_threads()));
    if (n_keys == 0) {
#pragma omp parallel for schedule(static) num_threads(T)
        for (int64_t i = 0; i < n; i++) out[i] = -1;
        return 0;
    }
    const int64_t lo = keys[0], hi = keys[n_keys - 1];
    const uint64_t span = (uint64_t)hi - (uint64_t)lo;
    if (span < (uint64_t)std::max<int64_t>(8 * n, (int64_t)1 << 22)) {
        std::vector<int32_t, NoInit<int32_t>> table((size_t)span + 1);
        int32_t *t = table.data();
        const bool small = n_keys < ((int64_t)1 << 31);
        if (small) {
#pragma omp parallel for schedule(static) num_threads(T)
            for (int64_t j = 0; j <= (int64_t)span; j++) t[j] = -1;
#pragma omp parallel for schedule(static) num_threads(T)
            for (int64_t j = 0; j < n_keys; j++) t[keys[j] - lo] = (int32_t)j;
#pragma omp parallel for schedule(static) num_threads(T)
            for (int64_t i = 0; i < n; i++) {
                const int64_t v = values[i];
                out[i] = (v < lo || v > hi) ? -1 : t[v - lo];
            }
            return 0;
        }
    }
#pragma omp parallel for schedule(static) num_threads(T)
    for (int64_t i = 0; i < n; i++) {
        const int64_t v = values[i];
        const int64_t *p = std::lower_bound(keys, keys + n_keys, v);
        out[i] = (p != keys + n_keys && *p == v) ? (int64_t)(p - keys) : -1;
    }
    return 0;
}

// out[i] = src[idx[i]] for elements of `elem` bytes (1, 4, 8 or 32): numpy's
// fancy-index gather on all threads.  Every idx[i] must lie in [0, n_src).
int taoamd_host_take(int32_t elem, int64_t n_src, const void *src, int64_t n,
                     const int64_t *idx, void *out)
{
    taoamd::ThreadScope threads;
    if (n < 0 || n_src < 0 || (n && (!src || !idx || !out))) return 1;
    const int T = std::max(1, std::min(32, taoamd::team_threads()));
    int bad = 0;
#define TAKE_LOOP(TYPE)                                                          \
    {                                                                             \
        const TYPE *s_ = (const TYPE *)src;                                       \
        TYPE *o_ = (TYPE *)out;                                                   \
        _Pragma("omp parallel for schedule(static) num_threads(T) reduction(| : bad)") \
        for (int64_t i = 0; i < n; i++) {                                         \
            const int64_t j = idx[i];                                             \
            if (j < 0 || j >= n_src) { bad |= 1; continue; }                      \
            o_[i] = s_[j];                                                        \
        }                                                                         \
    }
    struct B32 { uint64_t w[4]; };
    switch (elem) {
    case 1: TAKE_LOOP(uint8_t) break;
    case 4: TAKE_LOOP(uint32_t) break;
    case 8: TAKE_LOOP(uint64_t) break;
    case 32: TAKE_LOOP(B32) break;
    default: return 1;
    }
#undef TAKE_LOOP
    return bad ? 2 : 0;
}

// out[s] = (((0.0 + v[off[s]]) + v[off[s] + 1]) + ...) / len(s): Python's
// left-to-right ``sum(x['area'] for x in track) / len(track)`` per track
// (reference tao_amodal/tao.py:186-187); an empty segment gives 0.0 / 0 = NaN
// as numpy does.
int taoamd_host_seq_mean(int64_t n_seg, const int64_t *off, const double *vals,
                         double *out)
{
    taoamd::ThreadScope threads;
    if (n_seg < 0 || (n_seg && (!off || !out))) return 1;
    const int T = std::max(1, std::min(32, taoamd::team_threads()));
#pragma omp parallel for schedule(static, 256) num_threads(T)
    for (int64_t s = 0; s < n_seg; s++) {
        double acc = 0.0;
        for (int64_t i = off[s]; i < off[s + 1]; i++) acc = acc + vals[i];
        out[s] = acc / (double)(off[s + 1] - off[s]);
    }
    return 0;
}

// ``list(set(ids) & set(ids))`` of CPython 3.7 - 3.12 for ints 0 <= k < 2^61 - 1
// (hash(k) == k): the order in which the reference visits the images of the
// evaluated videos (tao_amodal/tao.py:224-230, img_ids = video_images).
// Objects/setobject.c: set(iterable) adds key by key (set_add_entry: linear
// probes of 9 neighbours, then i = 5 i + 1 + perturb; growth when
// fill * 5 >= mask * 3 to used * 4, used * 2 above 50000 entries, re-inserting
// in slot order); set_intersection() of two equally large sets walks the second
// operand in slot order and adds what the first contains to a NEW set the same
// way; list() walks that one in slot order.  (csrc/pyset.hpp restates the same
// text for the device; here on int64 keys, -1 = unused slot.)  Returns 0, or 3
// when a key lies outside the supported range (the caller uses the interpreter).
namespace {
struct IdSet {
    std::vector<int64_t, NoInit<int64_t>> t;
    uint64_t mask = 7, used = 0;
    IdSet() : t(8) { std::fill(t.begin(), t.end(), (int64_t)-1); }
    static void insert_clean(int64_t *t, uint64_t mask, int64_t key)
    {
        uint64_t perturb = (uint64_t)key, i = (uint64_t)key & mask;
        for (;;) {
            if (t[i] < 0) { t[i] = key; return; }
            if (i + 9 <= mask)
                for (int j = 1; j <= 9; j++)
                    if (t[i + j] < 0) { t[i + j] = key; return; }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    void resize(uint64_t minused)
    {
        uint64_t size = 8;
        while (size <= minused) size <<= 1;
        std::vector<int64_t, NoInit<int64_t>> n(size);
        std::fill(n.begin(), n.end(), (int64_t)-1);
        for (uint64_t i = 0; i <= mask; i++)
            if (t[i] >= 0) insert_clean(n.data(), size - 1, t[i]);
        t.swap(n);
        mask = size - 1;
    }
    void add(int64_t key)
    {
        uint64_t perturb = (uint64_t)key, i = (uint64_t)key & mask;
        for (;;) {
            uint64_t e = i;
            int probes = (i + 9 <= mask) ? 9 : 0;
            do {
                const int64_t c = t[e];
                if (c < 0) {
                    t[e] = key;
                    used++;
                    if (used * 5 >= mask * 3) resize(used > 50000 ? used * 2 : used * 4);
                    return;
                }
                if (c == key) return;
                e++;
            } while (probes--);
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
};
}  // namespace

int taoamd_host_pyset_self_and(int64_t n, const int64_t *ids, int64_t *out, int64_t *n_out)
{
    if (n < 0 || (n && (!ids || !out)) || !n_out) return 1;
    const int64_t limit = ((int64_t)1 << 61) - 1;
    for (int64_t i = 0; i < n; i++)
        if (ids[i] < 0 || ids[i] >= limit) return 3;
    IdSet b;
    for (int64_t i = 0; i < n; i++) b.add(ids[i]);
    IdSet r;
    for (uint64_t i = 0; i <= b.mask; i++)
        if (b.t[i] >= 0) r.add(b.t[i]);
    int64_t m = 0;
    for (uint64_t i = 0; i <= r.mask; i++)
        if (r.t[i] >= 0) out[m++] = r.t[i];
    *n_out = m;
    return 0;
}

// How many track ids occur with more than one video id (reference
// tools/eval_on_tao_amodal.py:44-58: the size of track_ids_to_update -- a track
// is in it iff some prediction's video differs from the FIRST one's, i.e. iff
// the id has two different videos, whatever the order).  Ids in a modest range
// through a dense table on all threads: a video per id from any of its rows
// (racing plain stores of equal-sized words: one of the stored values wins),
// then every row against it.  Returns 0; 3 when the ids span too wide a range
// (the caller's numpy statement takes over).
int taoamd_host_track_clash(int64_t n, const int64_t *tid, const int64_t *vid, int64_t *n_clash)
{
    taoamd::ThreadScope threads;
    if (n < 0 || (n && (!tid || !vid)) || !n_clash) return 1;
    *n_clash = 0;
    if (n == 0) return 0;
    const int T = std::max(1, std::min(32, taoamd::team_threads()));
    int64_t lo = INT64_MAX, hi = INT64_MIN;
#pragma omp parallel for schedule(static) num_threads(T) reduction(min : lo) reduction(max : hi)
    for (int64_t i = 0; i < n; i++) {
        lo = std::min(lo, tid[i]);
        hi = std::max(hi, tid[i]);
    }
    const uint64_t span = (uint64_t)hi - (uint64_t)lo;
    if (span >= (uint64_t)std::max<int64_t>(8 * n, (int64_t)1 << 22)) return 3;
    std::vector<int64_t, NoInit<int64_t>> of((size_t)span + 1);
    std::vector<uint8_t, NoInit<uint8_t>> bad((size_t)span + 1);
    int64_t *o = of.data();
    uint8_t *b = bad.data();
#pragma omp parallel for schedule(static) num_threads(T)
    for (int64_t j = 0; j <= (int64_t)span; j++) b[j] = 0;
#pragma omp parallel for schedule(static) num_threads(T)
    for (int64_t i = 0; i < n; i++) __atomic_store_n(&o[tid[i] - lo], vid[i], __ATOMIC_RELAXED);
#pragma omp parallel for schedule(static) num_threads(T)
    for (int64_t i = 0; i < n; i++)
        if (__atomic_load_n(&o[tid[i] - lo], __ATOMIC_RELAXED) != vid[i])
            __atomic_store_n(&b[tid[i] - lo], (uint8_t)1, __ATOMIC_RELAXED);
    int64_t c = 0;
#pragma omp parallel for schedule(static) num_threads(T) reduction(+ : c)
    for (int64_t j = 0; j <= (int64_t)span; j++) c += b[j];
    *n_clash = c;
    return 0;
}

// Boxes (x, y, w, h) with x < 0, y < 0, w <= 0 or h <= 0 -- the count behind the
// reference's "annotations had negative values in coordinates" warning
// (tao_amodal/tao.py:143-158) -- in one pass on all threads (numpy: four strided
// comparisons and three temporaries, 0.03 s for 1.5 M annotations).  NaN
// compares false, as in numpy.
int taoamd_host_count_bad_boxes(int64_t n, const double *bbox, int64_t *n_bad)
{
    taoamd::ThreadScope threads;
    if (n < 0 || (n && !bbox) || !n_bad) return 1;
    const int T = std::max(1, std::min(32, taoamd::team_threads()));
    int64_t c = 0;
#pragma omp parallel for schedule(static) num_threads(T) reduction(+ : c)
    for (int64_t i = 0; i < n; i++) {
        const double *b = bbox + 4 * i;
        c += (b[0] < 0) | (b[1] < 0) | (b[2] <= 0) | (b[3] <= 0);
    }
    *n_bad = c;
    return 0;
}

}  // extern "C"
