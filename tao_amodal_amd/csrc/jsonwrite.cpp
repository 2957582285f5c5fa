// Columns -> JSON text, the inverse of the native readers (ingest.cpp): the
// two input files of the evaluation path written from arrays.  Used where a
// synthetic set has to exist as FILES -- the wall-clock leg of bench.py and
// tools/wallclock_config2.py run the drop-in CLI on 30 M predictions, which
// json.dump of 30 M dicts cannot produce in reasonable time or memory.
//
// Numbers: integers as such, doubles by std::to_chars (shortest text that
// parses back to the same double, what Python's repr gives); an integral double
// is written without a fraction.  OpenMP: rows are formatted in blocks by all
// threads, blocks written in order.
#include "host_threads.hpp"
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tao_amodal_ingest.h"

namespace {

struct Out {
    std::string s;
    void lit(const char *t) { s.append(t); }
    void i64(int64_t v)
    {
        char b[24];
        auto r = std::to_chars(b, b + sizeof b, v);
        s.append(b, r.ptr);
    }
    void f64(double v)
    {
        if (std::isfinite(v) && v == std::floor(v) && std::fabs(v) < 9e15) {
            i64((int64_t)v);
            return;
        }
        char b[40];
        auto r = std::to_chars(b, b + sizeof b, v);
        s.append(b, r.ptr);
    }
    void list(const int64_t *v, int64_t a, int64_t b)
    {
        s.push_back('[');
        for (int64_t k = a; k < b; k++) {
            if (k > a) s.push_back(',');
            i64(v[k]);
        }
        s.push_back(']');
    }
};

// rows [0, n) formatted by `row(out, k)` in blocks, written in order
template <class Row>
bool write_rows(FILE *f, int64_t n, const Row &row)
{
    const int64_t block = 1 << 15;
    const int64_t wave = 64;                 // blocks formatted per round
    std::vector<std::string> parts(wave);
    for (int64_t b0 = 0; b0 * block < n; b0 += wave) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int64_t j = 0; j < wave; j++) {
            Out o;
            const int64_t lo = (b0 + j) * block, hi = std::min(n, lo + block);
            for (int64_t k = lo; k < hi; k++) {
                if (k) o.s.push_back(',');
                row(o, k);
            }
            parts[j].swap(o.s);
        }
        for (int64_t j = 0; j < wave; j++)
            if (!parts[j].empty() &&
                fwrite(parts[j].data(), 1, parts[j].size(), f) != parts[j].size())
                return false;
    }
    return true;
}

}  // namespace

extern "C" int taoamd_pred_write(const char *path, int64_t n, const int64_t *image_id,
                                 const int64_t *category_id, const double *bbox,
                                 const double *score, const int64_t *track_id,
                                 const int64_t *video_id)
{
    taoamd::ThreadScope threads;
    if (!path || n < 0 || (n && (!image_id || !category_id || !bbox || !score)))
        return 1;
    FILE *f = fopen(path, "wb");
    if (!f) return 2;
    bool ok = fputc('[', f) != EOF;
    ok = ok && write_rows(f, n, [&](Out &o, int64_t k) {
        o.lit("{\"image_id\":");
        o.i64(image_id[k]);
        o.lit(",\"category_id\":");
        o.i64(category_id[k]);
        o.lit(",\"bbox\":[");
        for (int j = 0; j < 4; j++) {
            if (j) o.s.push_back(',');
            o.f64(bbox[4 * k + j]);
        }
        o.lit("],\"score\":");
        {   // (a score stays a float in the file: 1.0, not 1)
            char b[40];
            auto r = std::to_chars(b, b + sizeof b, score[k]);
            o.s.append(b, r.ptr);
            if (score[k] == std::floor(score[k]) && !memchr(b, 'e', r.ptr - b) &&
                !memchr(b, '.', r.ptr - b))
                o.lit(".0");
        }
        if (track_id) {
            o.lit(",\"track_id\":");
            o.i64(track_id[k]);
        }
        if (video_id) {
            o.lit(",\"video_id\":");
            o.i64(video_id[k]);
        }
        o.s.push_back('}');
    });
    ok = ok && fputc(']', f) != EOF;
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : 3;
}

// arrays / counts in the order of GTColumns.FIELDS (tao_amodal_amd/columns.py)
enum {
    CAT_ID, CAT_FREQ, CAT_MERGED, VID_ID, VID_NEG_OFF, VID_NEG, VID_NEL_OFF, VID_NEL,
    IMG_ID, IMG_VID, IMG_FRAME, IMG_NEG_OFF, IMG_NEG, IMG_NEL_OFF, IMG_NEL,
    TRK_ID, TRK_CAT, TRK_VID, TRK_IGNORE,
    ANN_ID, ANN_IMG, ANN_TRK, ANN_CAT, ANN_BBOX, ANN_AREA, ANN_VIS, ANN_OOF, ANN_IGNORE,
    N_FIELDS
};

extern "C" int taoamd_gt_write(const char *path, const void *const *a, const int64_t *cnt,
                               int32_t n_fields)
{
    taoamd::ThreadScope threads;
    if (!path || !a || !cnt || n_fields != N_FIELDS) return 1;
    auto I = [&](int f) { return (const int64_t *)a[f]; };
    auto D = [&](int f) { return (const double *)a[f]; };
    auto B = [&](int f) { return (const uint8_t *)a[f]; };
    FILE *f = fopen(path, "wb");
    if (!f) return 2;
    bool ok = fputs("{\"info\":{\"description\":\"synthetic\"},\"images\":[", f) >= 0;
    ok = ok && write_rows(f, cnt[IMG_ID], [&](Out &o, int64_t k) {
        o.lit("{\"id\":");
        o.i64(I(IMG_ID)[k]);
        o.lit(",\"video_id\":");
        o.i64(I(IMG_VID)[k]);
        o.lit(",\"frame_index\":");
        o.f64(D(IMG_FRAME)[k]);
        o.lit(",\"neg_category_ids\":");
        o.list(I(IMG_NEG), I(IMG_NEG_OFF)[k], I(IMG_NEG_OFF)[k + 1]);
        o.lit(",\"not_exhaustive_category_ids\":");
        o.list(I(IMG_NEL), I(IMG_NEL_OFF)[k], I(IMG_NEL_OFF)[k + 1]);
        o.s.push_back('}');
    });
    ok = ok && fputs("],\"videos\":[", f) >= 0;
    ok = ok && write_rows(f, cnt[VID_ID], [&](Out &o, int64_t k) {
        o.lit("{\"id\":");
        o.i64(I(VID_ID)[k]);
        o.lit(",\"name\":\"v");
        o.i64(I(VID_ID)[k]);
        o.lit("\",\"neg_category_ids\":");
        o.list(I(VID_NEG), I(VID_NEG_OFF)[k], I(VID_NEG_OFF)[k + 1]);
        o.lit(",\"not_exhaustive_category_ids\":");
        o.list(I(VID_NEL), I(VID_NEL_OFF)[k], I(VID_NEL_OFF)[k + 1]);
        o.s.push_back('}');
    });
    ok = ok && fputs("],\"tracks\":[", f) >= 0;
    ok = ok && write_rows(f, cnt[TRK_ID], [&](Out &o, int64_t k) {
        o.lit("{\"id\":");
        o.i64(I(TRK_ID)[k]);
        o.lit(",\"category_id\":");
        o.i64(I(TRK_CAT)[k]);
        o.lit(",\"video_id\":");
        o.i64(I(TRK_VID)[k]);
        if (B(TRK_IGNORE)[k]) o.lit(",\"ignore\":1");
        o.s.push_back('}');
    });
    ok = ok && fputs("],\"annotations\":[", f) >= 0;
    ok = ok && write_rows(f, cnt[ANN_ID], [&](Out &o, int64_t k) {
        o.lit("{\"id\":");
        o.i64(I(ANN_ID)[k]);
        o.lit(",\"image_id\":");
        o.i64(I(ANN_IMG)[k]);
        o.lit(",\"track_id\":");
        o.i64(I(ANN_TRK)[k]);
        o.lit(",\"category_id\":");
        o.i64(I(ANN_CAT)[k]);
        o.lit(",\"bbox\":[");
        for (int j = 0; j < 4; j++) {
            if (j) o.s.push_back(',');
            o.f64(D(ANN_BBOX)[4 * k + j]);
        }
        o.lit("],\"area\":");
        o.f64(D(ANN_AREA)[k]);
        o.lit(",\"visibility\":");
        o.f64(D(ANN_VIS)[k]);
        o.lit(B(ANN_OOF)[k] ? ",\"out_of_frame\":true" : ",\"out_of_frame\":false");
        if (B(ANN_IGNORE)[k]) o.lit(",\"ignore\":1");
        o.s.push_back('}');
    });
    ok = ok && fputs("],\"categories\":[", f) >= 0;
    const int64_t n_merged = cnt[CAT_MERGED] / 2;
    ok = ok && write_rows(f, cnt[CAT_ID], [&](Out &o, int64_t k) {
        const int64_t id = I(CAT_ID)[k];
        o.lit("{\"id\":");
        o.i64(id);
        o.lit(",\"name\":\"c");
        o.i64(id);
        o.lit("\",\"frequency\":\"");
        o.s.push_back((char)B(CAT_FREQ)[k]);
        o.s.push_back('"');
        bool any = false;
        for (int64_t m = 0; m < n_merged; m++) {
            if (I(CAT_MERGED)[2 * m + 1] != id) continue;
            o.lit(any ? ",{\"id\":" : ",\"merged\":[{\"id\":");
            o.i64(I(CAT_MERGED)[2 * m]);
            o.s.push_back('}');
            any = true;
        }
        if (any) o.s.push_back(']');
        o.s.push_back('}');
    });
    ok = ok && fputs("]}", f) >= 0;
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : 3;
}
