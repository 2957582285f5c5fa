// Run-length masks on the host: what LVISEval(iou_type="segm") needs before
// any IoU is taken -- polygons rasterised and united, uncompressed and
// compressed RLE read, areas, tight boxes, the compressed text form.
//
// Replaces the calls the reference makes into pycocotools for this
//   LVIS.ann_to_rle                       lvis_amodal/lvis.py:171-193
//   mask_utils.area / toBbox              lvis_amodal/results.py:54-60
// i.e. frPyObjects / merge / area / toBbox of pycocotools' _mask.pyx over
// rleFrPoly, rleMerge, rleArea, rleToBbox, rleToString, rleFrString of its
// common/maskApi.c (vendored in the reference tree under
// visualization/tao/third_party/pysot/training_dataset/coco/pycocotools).
// Results are bit-identical to those (tests/test_masks.py: golden fixture F6
// and the reference C compiled in oracle/_ref).
//
// A mask = column-major run lengths starting with a (possibly empty) run of
// zeros.  A batch keeps its masks back to back (CSR), which is the layout the
// device kernel taoamd_rle_iou reads.
#include "host_threads.hpp"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tao_amodal_ingest.h"

namespace {

struct Mask {
    int64_t h = 0, w = 0;
    std::vector<uint32_t> runs;
};

struct Batch {
    std::vector<int64_t> off{0};
    std::vector<uint32_t> runs;
    std::vector<int32_t> hw;
    void push(const Mask &m)
    {
        runs.insert(runs.end(), m.runs.begin(), m.runs.end());
        off.push_back((int64_t)runs.size());
        hw.push_back((int32_t)m.h);
        hw.push_back((int32_t)m.w);
    }
    int64_t count() const { return (int64_t)off.size() - 1; }
};

// Points of the polygon outline on a grid five times finer than the pixels:
// every edge is walked one step at a time along its longer axis, the other
// coordinate rounded half up.  A repeated vertex contributes its own point.
void outline(const double *xy, int64_t k, std::vector<int> &us, std::vector<int> &vs)
{
    const double scale = 5;
    std::vector<int> px(k), py(k);
    for (int64_t j = 0; j < k; j++) {
        px[j] = (int)(scale * xy[2 * j] + .5);
        py[j] = (int)(scale * xy[2 * j + 1] + .5);
    }
    for (int64_t j = 0; j < k; j++) {
        int xs = px[j], ys = py[j], xe = px[(j + 1) % k], ye = py[(j + 1) % k];
        const int dx = std::abs(xe - xs), dy = std::abs(ys - ye);
        if (dx == 0 && dy == 0) {
            us.push_back(xs);
            vs.push_back(ys);
            continue;
        }
        const bool along_x = dx >= dy;
        const bool flip = along_x ? xs > xe : ys > ye;
        if (flip) {
            std::swap(xs, xe);
            std::swap(ys, ye);
        }
        if (along_x) {
            const double s = (double)(ye - ys) / dx;
            for (int d = 0; d <= dx; d++) {
                const int t = flip ? dx - d : d;
                us.push_back(t + xs);
                vs.push_back((int)(ys + s * t + .5));
            }
        } else {
            const double s = (double)(xe - xs) / dy;
            for (int d = 0; d <= dy; d++) {
                const int t = flip ? dy - d : d;
                vs.push_back(t + ys);
                us.push_back((int)(xs + s * t + .5));
            }
        }
    }
}

// One polygon -> runs.  Where the outline moves to another column a run
// boundary falls at (column, ceil(row)) in pixel units; boundaries outside
// whole columns of the frame are dropped, rows are clamped to [0, h].  The
// boundaries, as column-major offsets and sorted, delimit the runs; empty
// runs are folded away.
Mask from_polygon(const double *xy, int64_t k, int64_t h, int64_t w)
{
    std::vector<int> us, vs;
    outline(xy, k, us, vs);
    const double scale = 5;
    std::vector<uint32_t> cut;
    for (size_t j = 1; j < us.size(); j++) {
        if (us[j] == us[j - 1]) continue;
        double xd = (double)(us[j] < us[j - 1] ? us[j] : us[j] - 1);
        xd = (xd + .5) / scale - .5;
        if (std::floor(xd) != xd || xd < 0 || xd > (double)(w - 1)) continue;
        double yd = (double)std::min(vs[j], vs[j - 1]);
        yd = (yd + .5) / scale - .5;
        if (yd < 0) yd = 0; else if (yd > (double)h) yd = (double)h;
        yd = std::ceil(yd);
        cut.push_back((uint32_t)((int)xd * (int)h + (int)yd));
    }
    cut.push_back((uint32_t)(h * w));
    std::sort(cut.begin(), cut.end());
    Mask m;
    m.h = h;
    m.w = w;
    uint32_t prev = 0;
    bool glue = false;          // the next run extends the last one
    for (size_t j = 0; j < cut.size(); j++) {
        const uint32_t len = cut[j] - prev;
        prev = cut[j];
        if (j == 0) {
            m.runs.push_back(len);
        } else if (glue) {
            m.runs.back() += len;
            glue = false;
        } else if (len > 0) {
            m.runs.push_back(len);
        } else {
            glue = true;
        }
    }
    return m;
}

// Cursor over the runs of a mask.
struct Cursor {
    const uint32_t *c;
    size_t n, i = 1;
    uint32_t left;
    int v = 0;
    Cursor(const uint32_t *c_, size_t n_) : c(c_), n(n_), left(c_[0]) {}
    void take(uint32_t k)
    {
        left -= k;
        if (left == 0 && i < n) {
            left = c[i++];
            v ^= 1;
        }
    }
};

// Union of masks of one frame size, folded left to right.
Mask unite(const std::vector<Mask> &parts)
{
    Mask acc;
    if (parts.empty()) return acc;
    acc = parts[0];
    for (size_t p = 1; p < parts.size(); p++) {
        const Mask &o = parts[p];
        if (o.h != acc.h || o.w != acc.w) return Mask();
        if (acc.runs.empty() || o.runs.empty()) return Mask();
        Cursor a(acc.runs.data(), acc.runs.size()), b(o.runs.data(), o.runs.size());
        std::vector<uint32_t> out;
        uint32_t run = 0;
        int v = 0;
        for (;;) {
            const uint32_t k = std::min(a.left, b.left);
            run += k;
            a.take(k);
            b.take(k);
            const uint32_t rest = a.left + b.left;
            const int nv = a.v | b.v;
            if (nv != v || rest == 0) {
                out.push_back(run);
                run = 0;
            }
            v = nv;
            if (rest == 0) break;
        }
        acc.runs.swap(out);
    }
    return acc;
}

// The text form: 5 data bits and a continuation bit per character, offset
// 48; from the fourth run on the difference to the run two places back.
Mask from_text(const char *s, int64_t h, int64_t w)
{
    Mask m;
    m.h = h;
    m.w = w;
    size_t p = 0;
    while (s[p]) {
        long x = 0;
        int k = 0;
        for (;;) {
            const int ch = s[p] - 48;
            x |= (long)(ch & 0x1f) << (5 * k);
            p++;
            k++;
            if (!(ch & 0x20)) {
                if (ch & 0x10) x |= -1L << (5 * k);
                break;
            }
            if (!s[p]) break;       // truncated text: stop, do not run past it
        }
        if (m.runs.size() > 2) x += (long)m.runs[m.runs.size() - 2];
        m.runs.push_back((uint32_t)x);
    }
    return m;
}

std::string to_text(const uint32_t *c, int64_t n)
{
    std::string s;
    for (int64_t i = 0; i < n; i++) {
        long x = (long)c[i];
        if (i > 2) x -= (long)c[i - 2];
        for (;;) {
            int ch = (int)(x & 0x1f);
            x >>= 5;
            const bool more = (ch & 0x10) ? x != -1 : x != 0;
            if (more) ch |= 0x20;
            s.push_back((char)(ch + 48));
            if (!more) break;
        }
    }
    return s;
}

}  // namespace

extern "C" {

void *taoamd_rle_new(void) { return new Batch(); }

void taoamd_rle_free(void *handle) { delete (Batch *)handle; }

int64_t taoamd_rle_count(void *handle) { return ((Batch *)handle)->count(); }

int64_t taoamd_rle_total(void *handle)
{
    return (int64_t)((Batch *)handle)->runs.size();
}

int64_t taoamd_rle_add_polygons(void *handle, int32_t n_parts,
                                const int64_t *part_off, const double *xy,
                                int64_t height, int64_t width)
{
    Batch *b = (Batch *)handle;
    if (n_parts < 1 || height < 0 || width < 0 ||
        (uint64_t)height * (uint64_t)width >= (1ull << 32))
        return -1;
    std::vector<Mask> parts;
    for (int32_t p = 0; p < n_parts; p++) {
        const int64_t len = part_off[p + 1] - part_off[p];
        if (len < 2) return -1;
        parts.push_back(from_polygon(xy + part_off[p], len / 2, height, width));
    }
    b->push(unite(parts));
    return b->count() - 1;
}

int64_t taoamd_rle_add_polygon_batch(void *handle, int64_t n_masks,
                                     const int64_t *mask_part_off,
                                     const int64_t *part_off, const double *xy,
                                     const int32_t *hw)
{
    taoamd::ThreadScope threads;
    Batch *b = (Batch *)handle;
    if (n_masks < 0) return -1;
    for (int64_t m = 0; m < n_masks; m++) {
        const int64_t h = hw[2 * m], w = hw[2 * m + 1];
        if (mask_part_off[m + 1] <= mask_part_off[m] || h < 0 || w < 0 ||
            (uint64_t)h * (uint64_t)w >= (1ull << 32))
            return -1;
        for (int64_t p = mask_part_off[m]; p < mask_part_off[m + 1]; p++)
            if (part_off[p + 1] - part_off[p] < 2) return -1;
    }
    // the masks are independent: rasterise them on all cores, append in order
    std::vector<Mask> made((size_t)n_masks);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t m = 0; m < n_masks; m++) {
        std::vector<Mask> parts;
        for (int64_t p = mask_part_off[m]; p < mask_part_off[m + 1]; p++)
            parts.push_back(from_polygon(xy + part_off[p],
                                         (part_off[p + 1] - part_off[p]) / 2,
                                         hw[2 * m], hw[2 * m + 1]));
        made[(size_t)m] = unite(parts);
    }
    const int64_t first = b->count();
    for (const Mask &m : made) b->push(m);
    return first;
}

int64_t taoamd_rle_add_counts(void *handle, const uint32_t *counts, int64_t m,
                              int64_t height, int64_t width)
{
    Batch *b = (Batch *)handle;
    if (m < 0 || height < 0 || width < 0) return -1;
    Mask k;
    k.h = height;
    k.w = width;
    k.runs.assign(counts, counts + m);
    b->push(k);
    return b->count() - 1;
}

int64_t taoamd_rle_add_string(void *handle, const char *text, int64_t height,
                              int64_t width)
{
    Batch *b = (Batch *)handle;
    if (!text || height < 0 || width < 0) return -1;
    b->push(from_text(text, height, width));
    return b->count() - 1;
}

int taoamd_rle_copy(void *handle, int64_t *off, uint32_t *counts, int32_t *hw,
                    uint32_t *area, double *bbox)
{
    Batch *b = (Batch *)handle;
    const int64_t n = b->count();
    if (off) std::memcpy(off, b->off.data(), (n + 1) * sizeof(int64_t));
    if (counts && !b->runs.empty())
        std::memcpy(counts, b->runs.data(), b->runs.size() * sizeof(uint32_t));
    if (hw && n) std::memcpy(hw, b->hw.data(), 2 * n * sizeof(int32_t));
    for (int64_t i = 0; i < n; i++) {
        const uint32_t *c = b->runs.data() + b->off[i];
        const int64_t m = b->off[i + 1] - b->off[i];
        if (area) {
            uint32_t a = 0;
            for (int64_t j = 1; j < m; j += 2) a += c[j];
            area[i] = a;
        }
        if (bbox) {
            // tight box of the runs' end points; an odd last run (trailing
            // zeros) is not looked at, no run at all gives four zeros
            const uint32_t h = (uint32_t)b->hw[2 * i], w = (uint32_t)b->hw[2 * i + 1];
            const int64_t me = (m / 2) * 2;
            double *o = bbox + 4 * i;
            if (me == 0 || h == 0) {
                o[0] = o[1] = o[2] = o[3] = 0;
                continue;
            }
            uint32_t xs = w, ys = h, xe = 0, ye = 0, cc = 0;
            for (int64_t j = 0; j < me; j++) {
                cc += c[j];
                const uint32_t t = cc - (uint32_t)(j % 2);
                const uint32_t y = t % h, x = (t - y) / h;
                xs = std::min(xs, x);
                xe = std::max(xe, x);
                ys = std::min(ys, y);
                ye = std::max(ye, y);
            }
            o[0] = xs;
            o[2] = (uint32_t)(xe - xs + 1);
            o[1] = ys;
            o[3] = (uint32_t)(ye - ys + 1);
        }
    }
    return 0;
}

int64_t taoamd_rle_string(void *handle, int64_t index, char *buf, int64_t cap)
{
    Batch *b = (Batch *)handle;
    if (index < 0 || index >= b->count()) return -1;
    const std::string s =
        to_text(b->runs.data() + b->off[index], b->off[index + 1] - b->off[index]);
    if (buf && cap > 0) {
        const int64_t k = std::min<int64_t>(cap - 1, (int64_t)s.size());
        std::memcpy(buf, s.data(), k);
        buf[k] = 0;
    }
    return (int64_t)s.size();
}

}  // extern "C"
