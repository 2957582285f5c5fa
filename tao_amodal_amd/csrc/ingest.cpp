// Native columnar ingest of the prediction list (host only, no GPU code).
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
// Numbers are converted with std::from_chars (correctly rounded, the same
// value Python's float() gives), so the columns are bit-identical to what
// `json.load` + numpy conversion produce (tests/test_ingest.py).
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Columns {
    std::vector<int64_t> image_id, category_id, track_id, video_id;
    std::vector<double> bbox, score;
    std::string error;
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

bool key_is(const char *b, const char *e, const char *name)
{
    size_t n = strlen(name);
    return (size_t)(e - b) == n && !memcmp(b, name, n);
}

// one prediction object [b, e) -> row i of the columns
bool parse_object(const char *b, const char *e, int64_t i, Columns &c, std::string &err)
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
            if (key_is(kb, ke, "image_id")) { if (cur.integer(iv)) { c.image_id[i] = iv; has_img = true; } }
            else if (key_is(kb, ke, "category_id")) { if (cur.integer(iv)) { c.category_id[i] = iv; has_cat = true; } }
            else if (key_is(kb, ke, "track_id")) { if (cur.integer(iv)) c.track_id[i] = iv; }
            else if (key_is(kb, ke, "video_id")) { if (cur.integer(iv)) c.video_id[i] = iv; }
            else if (key_is(kb, ke, "score")) { if (cur.number(v)) { c.score[i] = v; has_score = true; } }
            else if (key_is(kb, ke, "bbox")) {
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

}  // namespace

extern "C" {

void *taoamd_pred_parse(const char *path, char *err, size_t errlen)
{
    auto fail = [&](const std::string &m) -> void * {
        if (err && errlen) snprintf(err, errlen, "%s", m.c_str());
        return nullptr;
    };
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(std::string("cannot open ") + path);
    struct stat st;
    fstat(fd, &st);
    size_t len = (size_t)st.st_size;
    const char *buf = len ? (const char *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0) : "";
    close(fd);
    if (len && buf == MAP_FAILED) return fail("mmap failed");
    const char *p = buf, *e = buf + len;
    auto done = [&]() { if (len) munmap((void *)buf, len); };
    while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++;
    if (p >= e || *p != '[') { done(); return fail("results is not a list."); }
    p++;
    // pass 1: object boundaries at depth 1
    std::vector<std::pair<size_t, size_t>> objs;
    {
        int depth = 0;
        size_t start = 0;
        bool closed = false;
        while (p < e) {
            char c = *p;
            if (c == '"') {
                p++;
                while (p < e && *p != '"') { if (*p == '\\') p++; p++; }
                p++;
                continue;
            }
            if (c == '{' || c == '[') { if (depth == 0) start = (size_t)(p - buf); depth++; }
            else if (c == '}' || c == ']') {
                if (depth == 0) { closed = c == ']'; break; }
                depth--;
                if (depth == 0) objs.emplace_back(start, (size_t)(p - buf) + 1);
            }
            p++;
        }
        if (!closed) { done(); return fail("unterminated list"); }
    }
    Columns *c = new Columns;
    const int64_t n = (int64_t)objs.size();
    c->image_id.resize(n); c->category_id.resize(n); c->track_id.resize(n);
    c->video_id.resize(n); c->score.resize(n); c->bbox.resize(4 * n);
    bool ok = true;
    std::string first_err;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        std::string er;
        if (buf[objs[i].first] != '{') er = "list element is not an object";
        if (!er.empty() || !parse_object(buf + objs[i].first, buf + objs[i].second, i, *c, er)) {
#pragma omp critical
            { if (ok) { ok = false; first_err = "prediction " + std::to_string(i) + ": " + er; } }
        }
    }
    done();
    if (!ok) { delete c; return fail(first_err); }
    return c;
}

int64_t taoamd_pred_count(void *h) { return (int64_t)((Columns *)h)->image_id.size(); }

int taoamd_pred_copy(void *h, int64_t *image_id, int64_t *category_id, double *bbox,
                     double *score, int64_t *track_id, int64_t *video_id)
{
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

}  // extern "C"
