// The prediction file read ON THE DEVICE (gfx950): prediction.json -- 3.7 GB of
// text for 30 M boxes at the validation scale -- is copied into HBM as it is and
// turned into the six columns of DTColumns there.  The reference json.load()s the
// file into ~N dicts, twice (lvis_amodal/results.py:29-30, tools/
// eval_on_tao_amodal.py:127-128); the host reader of csrc/ingest.cpp takes
// 0.6-0.7 s of sixteen cores for it, most of the drop-in CLI's wall-clock.  This
// is byte work at HBM speed: every pass streams the text once.
//
//   js_quotes   per 16 KB block: parity of the quote count, any backslash
//   js_depth    with the blocks' string state known: bracket depth change
//   js_count    with the blocks' depth known: top-level objects per block, and
//               the checks of the list's shape (what may stand between objects)
//   js_starts   byte offset of every top-level object, in file order
//   js_parse    one thread per object: the known keys' numbers, converted with
//               the correctly rounded decimal -> double of decfloat.hpp
//   (three single-workgroup scans over the per-block tables in between)
//
// The device path is OPTIMISTIC.  Whatever it is not sure to read exactly as
// json.load + the reference would -- a backslash anywhere in the file, anything
// but commas and white space between the list's objects, and per object: an
// unknown literal (true / NaN / Infinity), a number of more than 19 digits, an id
// that is not a plain integer, a missing key, any syntax it does not expect -- is
// LEFT TO THE HOST READER: the whole file (status 1), or the objects it lists
// (their rows are then parsed by csrc/ingest.cpp's parse_object and patched in,
// with that reader's error messages).  Nothing is guessed.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "common.hpp"
#include "decfloat.hpp"

using namespace taoamd;

#define JS_T 256                  // threads of a block
#define JS_W 16                   // 32-bit words of text per thread
#define JS_B (4 * JS_W)           // bytes per thread
#define JS_BLK (JS_T * JS_B)      // bytes per block (16 KB)

#define JS_BAD_BACKSLASH 1u       // bits of the anomaly word
#define JS_BAD_SHAPE 2u

struct JsArgs {
    const uint32_t *text;         // padded with blanks to whole blocks
    int64_t len;                  // bytes of the file
    int32_t n_blk;
    int32_t *blk_par, *blk_delta, *blk_count;      // per block
    const int32_t *par_ex, *depth_ex, *count_ex;   // their exclusive prefix sums
    uint32_t *bad;                // anomaly bits
    uint32_t *n_open0;            // '[' at depth 0
    int64_t *starts;              // [n objects]
};

__device__ __forceinline__ uint32_t bytes_equal(uint32_t w, uint32_t c)
{
    // 0x80 in every byte of w that equals c (exact: no borrow between bytes)
    const uint32_t x = w ^ (c * 0x01010101u);
    return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
}

__device__ __forceinline__ void js_load(const JsArgs &a, uint32_t (&w)[JS_W])
{
    const uint4 *p = reinterpret_cast<const uint4 *>(a.text) +
                     ((int64_t)blockIdx.x * JS_T + threadIdx.x) * (JS_W / 4);
#pragma unroll
    for (int k = 0; k < JS_W / 4; k++) {
        const uint4 v = p[k];
        w[4 * k] = v.x;
        w[4 * k + 1] = v.y;
        w[4 * k + 2] = v.z;
        w[4 * k + 3] = v.w;
    }
}

// exclusive prefix sum of one value per thread over the block's JS_T threads
// (+ the block's total)
__device__ __forceinline__ int32_t block_scan(int32_t v, int32_t &total)
{
    __shared__ int32_t s_w[JS_T / WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t x = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const int32_t y = __shfl_up(x, d, WAVE);
        if (lane >= d) x += y;
    }
    __syncthreads();              // (s_w of an earlier call has been read)
    if (lane == WAVE - 1) s_w[wave] = x;
    __syncthreads();
    int32_t before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < JS_T / WAVE; k++) {
        const int32_t t = s_w[k];
        before += k < wave ? t : 0;
        all += t;
    }
    total = all;
    return before + x - v;
}

__device__ __forceinline__ int32_t quotes_of(const uint32_t (&w)[JS_W])
{
    int32_t q = 0;
#pragma unroll
    for (int k = 0; k < JS_W; k++) q += __popc(bytes_equal(w[k], '"'));
    return q;
}

__global__ __launch_bounds__(JS_T) void js_quotes_kernel(JsArgs a)
{
    uint32_t w[JS_W];
    js_load(a, w);
    uint32_t bs = 0;
#pragma unroll
    for (int k = 0; k < JS_W; k++) bs |= bytes_equal(w[k], '\\');
    if (bs) atomicOr(a.bad, JS_BAD_BACKSLASH);
    int32_t total;
    block_scan(quotes_of(w), total);
    if (threadIdx.x == 0) a.blk_par[blockIdx.x] = total & 1;
}

// One walk over a thread's bytes.  in_str / depth: the state before its first
// byte.  MODE 0: depth change only; 1: count the objects that start here and
// check the list's shape; 2: write the objects' offsets from out[0] on.
template <int MODE>
__device__ __forceinline__ int32_t js_walk(const uint32_t (&w)[JS_W], bool in_str, int32_t depth,
                                           int64_t byte0, int64_t len, uint32_t &bad,
                                           uint32_t &open0, int64_t *out)
{
    int32_t d = depth, n = 0;
#pragma unroll
    for (int k = 0; k < JS_W; k++) {
        // (a word without quote or bracket outside a string changes nothing
        // but the shape check: most words of a number-heavy file)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t c = (w[k] >> (8 * b)) & 0xffu;
            if (c == '"') {
                if (MODE == 1 && !in_str && d <= 1) bad |= JS_BAD_SHAPE;     // a string in the list
                in_str = !in_str;
            } else if (!in_str) {
                if (c == '{' || c == '[') {
                    if (MODE == 1) {
                        if (d == 0) {
                            if (c == '[') open0++; else bad |= JS_BAD_SHAPE;
                        } else if (d == 1 && c == '[') bad |= JS_BAD_SHAPE;
                    }
                    if (d == 1 && c == '{') {
                        if (MODE == 2) out[n] = byte0 + 4 * k + b;
                        n++;
                    }
                    d++;
                } else if (c == '}' || c == ']') {
                    d--;
                    if (MODE == 1 && d < 0) bad |= JS_BAD_SHAPE;
                } else if (MODE == 1 && d <= 1) {
                    const bool ws = c == ' ' || c == '\n' || c == '\t' || c == '\r';
                    if (!ws && !(d == 1 && c == ',') && byte0 + 4 * k + b < len)
                        bad |= JS_BAD_SHAPE;
                }
            }
        }
    }
    return MODE == 0 ? d - depth : n;
}

template <int MODE>
__global__ __launch_bounds__(JS_T) void js_walk_kernel(JsArgs a)
{
    uint32_t w[JS_W];
    js_load(a, w);
    const int64_t byte0 = ((int64_t)blockIdx.x * JS_T + threadIdx.x) * JS_B;
    int32_t total;
    const int32_t qb = block_scan(quotes_of(w), total);
    const bool in_str = ((a.par_ex[blockIdx.x] + qb) & 1) != 0;
    uint32_t bad = 0, open0 = 0;
    const int32_t delta = js_walk<0>(w, in_str, 0, byte0, a.len, bad, open0, nullptr);
    const int32_t db = block_scan(delta, total);
    if (MODE == 0) {
        if (threadIdx.x == 0) a.blk_delta[blockIdx.x] = total;
        return;
    }
    const int32_t depth = a.depth_ex[blockIdx.x] + db;
    const int32_t n = js_walk<1>(w, in_str, depth, byte0, a.len, bad, open0, nullptr);
    const int32_t nb = block_scan(n, total);
    if (MODE == 1) {
        if (threadIdx.x == 0) a.blk_count[blockIdx.x] = total;
        if (bad) atomicOr(a.bad, bad);
        if (open0) atomicAdd(a.n_open0, open0);
        return;
    }
    if (n) js_walk<2>(w, in_str, depth, byte0, a.len, bad, open0,
                      a.starts + a.count_ex[blockIdx.x] + nb);
}

// exclusive prefix sums of a per-block table by ONE workgroup (a table has
// 16 K entries per GB of text); out[n] = the total
__global__ __launch_bounds__(1024) void js_scan_kernel(const int32_t *__restrict__ in,
                                                       int32_t *__restrict__ out, int32_t n)
{
    __shared__ int32_t s_w[16];
    __shared__ int32_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int32_t base = 0; base < n; base += 1024) {
        const int32_t i = base + threadIdx.x;
        const int32_t v = i < n ? in[i] : 0;
        int32_t x = v;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const int32_t y = __shfl_up(x, d, WAVE);
            if (lane >= d) x += y;
        }
        if (lane == WAVE - 1) s_w[wave] = x;
        __syncthreads();
        int32_t before = s_carry;
        for (int k = 0; k < wave; k++) before += s_w[k];
        if (i < n) out[i] = before + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = s_carry;
}

// ---------------------------------------------------------------- objects
struct JsCols {
    int64_t *image_id, *category_id, *track_id, *video_id;
    double *bbox, *score;
};

struct JsParseArgs {
    const uint8_t *text;
    int64_t len, n;
    const int64_t *starts;
    JsCols c;
    int32_t *n_flag;              // objects left to the host reader
    int64_t *flag, *flag_at;      // their numbers and byte offsets (the first flag_cap of them)
    int32_t flag_cap;
};

__device__ __forceinline__ bool js_ws(uint32_t c)
{
    return c == ' ' || c == '\n' || c == '\t' || c == '\r';
}
__device__ __forceinline__ bool js_delim(uint32_t c)
{
    return c == ',' || c == '}' || c == ']' || js_ws(c);
}

// key [kb, ke) == name?
__device__ __forceinline__ bool js_key(const uint8_t *t, int64_t kb, int64_t ke, const char *name,
                                       int n)
{
    if (ke - kb != n) return false;
    for (int i = 0; i < n; i++)
        if (t[kb + i] != (uint8_t)name[i]) return false;
    return true;
}

// One prediction object: csrc/ingest.cpp's parse_object, minus everything it
// does for inputs that are not plain (those answer false: the host reads them).
__device__ bool js_object(const JsParseArgs &a, int64_t i)
{
    const uint8_t *t = a.text;
    int64_t p = a.starts[i] + 1;                       // behind the '{'
    const int64_t e = i + 1 < a.n ? a.starts[i + 1] : a.len;
    auto ws = [&]() { while (p < e && js_ws(t[p])) p++; };
    auto number = [&](double &v, bool *is_int, int64_t *iv) {
        const int64_t avail = e - p;
        const int n = avail > 64 ? 64 : (int)avail;
        const int used = decf::parse_json_number([&](int k) { return (unsigned)t[p + k]; }, n, v,
                                                 is_int, iv);
        if (used == 0 || (used == n && n == 64)) return false;
        p += used;
        return p < e && js_delim(t[p]);
    };
    bool has_img = false, has_cat = false, has_box = false, has_score = false;
    int64_t img = 0, cat = 0, trk = -1, vid = -1;
    double box[4] = {0, 0, 0, 0}, score = 0;
    ws();
    if (p < e && t[p] == '}') return false;            // (no keys: KeyError on the host)
    for (;;) {
        ws();
        if (p >= e || t[p] != '"') return false;
        const int64_t kb = ++p;
        while (p < e && t[p] != '"') p++;              // (no backslash in the file)
        if (p >= e) return false;
        const int64_t ke = p++;
        ws();
        if (p >= e || t[p] != ':') return false;
        p++;
        ws();
        if (p >= e) return false;
        double v;
        bool is_int;
        int64_t iv;
        if (js_key(t, kb, ke, "image_id", 8)) {
            if (!number(v, &is_int, &iv) || !is_int) return false;
            img = iv;
            has_img = true;
        } else if (js_key(t, kb, ke, "category_id", 11)) {
            if (!number(v, &is_int, &iv) || !is_int) return false;
            cat = iv;
            has_cat = true;
        } else if (js_key(t, kb, ke, "track_id", 8)) {
            if (!number(v, &is_int, &iv) || !is_int) return false;
            trk = iv;
        } else if (js_key(t, kb, ke, "video_id", 8)) {
            if (!number(v, &is_int, &iv) || !is_int) return false;
            vid = iv;
        } else if (js_key(t, kb, ke, "score", 5)) {
            if (!number(v, nullptr, nullptr)) return false;     // (true / NaN / ...: the host)
            score = v;
            has_score = true;
        } else if (js_key(t, kb, ke, "bbox", 4)) {
            if (t[p] != '[') return false;
            p++;
            for (int k = 0; k < 4; k++) {
                ws();
                if (k) {
                    if (p >= e || t[p] != ',') return false;
                    p++;
                    ws();
                }
                if (!number(v, nullptr, nullptr)) return false;
                box[k] = v;
            }
            ws();
            if (p >= e || t[p] != ']') return false;
            p++;
            has_box = true;
        } else {
            // any other value is skipped
            const uint32_t c = t[p];
            if (c == '"') {
                p++;
                while (p < e && t[p] != '"') p++;
                if (p >= e) return false;
                p++;
            } else if (c == '{' || c == '[') {
                int depth = 0;
                bool in = false;
                for (; p < e; p++) {
                    const uint32_t d = t[p];
                    if (d == '"') in = !in;
                    else if (!in) {
                        if (d == '{' || d == '[') depth++;
                        else if (d == '}' || d == ']') {
                            if (--depth == 0) break;
                        }
                    }
                }
                if (p >= e) return false;
                p++;
            } else {
                while (p < e && !js_delim(t[p])) p++;
            }
        }
        ws();
        if (p >= e) return false;
        if (t[p] == ',') {
            p++;
            continue;
        }
        if (t[p] == '}') break;
        return false;
    }
    if (!(has_img && has_cat && has_box && has_score)) return false;
    a.c.image_id[i] = img;
    a.c.category_id[i] = cat;
    a.c.track_id[i] = trk;
    a.c.video_id[i] = vid;
    a.c.score[i] = score;
    reinterpret_cast<double4 *>(a.c.bbox)[i] = make_double4(box[0], box[1], box[2], box[3]);
    return true;
}

__global__ __launch_bounds__(256) void js_parse_kernel(JsParseArgs a)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    if (!js_object(a, i)) {
        const int32_t k = atomicAdd(a.n_flag, 1);
        if (k < a.flag_cap) {
            a.flag[k] = i;
            a.flag_at[k] = a.starts[i];
        }
    }
}

// ------------------------------------------------------------------- host
namespace {

struct JsHandle {
    int fd = -1;
    const char *map = nullptr;
    size_t len = 0;
    uint8_t *d_text = nullptr, *d_cols = nullptr;
    int64_t *d_starts = nullptr, *d_flag = nullptr;
    bool own_text = true, own_starts = true;      // false: inside the caller's workspace
    uint8_t *work = nullptr;                      // the caller's workspace, what is left of it
    size_t work_left = 0;
    int64_t n = 0;
    hipStream_t s = nullptr;
    // n bytes of the caller's workspace (256-byte aligned), nullptr when it has not got them
    uint8_t *carve(size_t n_bytes)
    {
        const size_t need = (n_bytes + 255) & ~(size_t)255;
        if (!work || work_left < need) return nullptr;
        uint8_t *p = work;
        work += need;
        work_left -= need;
        return p;
    }
    ~JsHandle()
    {
        if (d_text && own_text) (void)hipFree(d_text);
        if (d_cols) (void)hipFree(d_cols);
        if (d_starts && own_starts) (void)hipFree(d_starts);
        if (map && len) munmap((void *)map, len);
    }
};

void js_msg(char *err, size_t errlen, const std::string &m)
{
    if (err && errlen) snprintf(err, errlen, "%s", m.c_str());
}

// Copies between PAGEABLE host memory and the device by several threads, each
// with a stream of its own: a pageable copy is staged through the runtime's
// pinned buffers by the calling thread's memcpy (and, for memory touched for the
// first time -- a mapped file, a fresh numpy array -- its page faults): one
// thread moves 14-38 GB/s, the link takes more.
struct Piece {
    char *dst;
    const char *src;
    size_t bytes;
};
hipError_t parallel_copy(const std::vector<Piece> &parts, hipMemcpyKind kind, int n_threads)
{
    const size_t SL = (size_t)64 << 20;
    std::vector<Piece> slices;
    for (const Piece &p : parts)
        for (size_t off = 0; off < p.bytes; off += SL)
            slices.push_back(Piece{p.dst + off, p.src + off, p.bytes - off < SL ? p.bytes - off : SL});
    std::atomic<size_t> next{0};
    std::atomic<int> failed{(int)hipSuccess};
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto work = [&]() {
        (void)hipSetDevice(dev);
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= slices.size()) break;
            const Piece &p = slices[k];
            hipError_t e = hipMemcpyAsync(p.dst, p.src, p.bytes, kind, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) failed.store((int)e);
        }
        if (st) (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> pool;
    const int nt = n_threads < (int)slices.size() ? n_threads : (int)slices.size();
    for (int t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    return (hipError_t)failed.load();
}

// Pages of a host range touched by helper threads while something else runs:
// a mapped file's pages are mapped on first read, a fresh array's on first
// write (huge pages where the system hands them out on request) -- faults the
// runtime's staging memcpy would otherwise take one by one on the copying thread.
struct Toucher {
    std::vector<std::thread> pool;
    void read(const char *p, size_t bytes, int n)
    {
        for (int w = 0; w < n; w++)
            pool.emplace_back([=]() {
                volatile char sink = 0;
                const size_t a = bytes * w / n, b = bytes * (w + 1) / n;
                for (size_t q = a; q < b; q += 4096) sink += p[q];
                (void)sink;
            });
    }
    void write(char *p, size_t bytes, int n)
    {
        const uintptr_t a0 = ((uintptr_t)p + ((size_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1);
        const uintptr_t b0 = ((uintptr_t)p + bytes) & ~(((uintptr_t)2 << 20) - 1);
        if (b0 > a0) madvise((void *)a0, b0 - a0, MADV_HUGEPAGE);
        for (int w = 0; w < n; w++)
            pool.emplace_back([=]() {
                const size_t a = bytes * w / n, b = bytes * (w + 1) / n;
                for (size_t q = a; q < b; q += 4096) p[q] = 0;
            });
    }
    void join()
    {
        for (auto &t : pool) t.join();
        pool.clear();
    }
    ~Toucher() { join(); }
};

int copy_threads()
{
    const char *e = getenv("TAOAMD_INGEST_COPY_THREADS");
    const int n = e ? atoi(e) : 6;
    return n < 1 ? 1 : n > 32 ? 32 : n;
}

}  // namespace

// (no room for the text or a table in HBM: the host reader's file, not an error)
#define JS_ALLOC(call)                                                 \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ == hipErrorOutOfMemory || e_ == hipErrorMemoryAllocation) { \
            (void)hipGetLastError();                                   \
            *status = TAOAMD_JSON_FALLBACK;                            \
            return nullptr;                                            \
        }                                                              \
        if (e_ != hipSuccess) {                                        \
            taoamd::set_error(e_, #call);                              \
            js_msg(err, errlen, std::string("HIP: ") + #call);         \
            *status = TAOAMD_ERR_HIP;                                  \
            return nullptr;                                            \
        }                                                              \
    } while (0)

#define JS_HIP(call)                                                   \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) {                                        \
            taoamd::set_error(e_, #call);                              \
            js_msg(err, errlen, std::string("HIP: ") + #call);         \
            *status = TAOAMD_ERR_HIP;                                  \
            return nullptr;                                            \
        }                                                              \
    } while (0)

// status: 0 ok; TAOAMD_ERR_HIP; TAOAMD_ERR_ARG (cannot open: err says why);
// TAOAMD_JSON_FALLBACK = the host reader should take the file
// Device memory the reader wants for a file of file_bytes (text, per-block
// tables, scratch, the objects' offsets for up to one object per 32 bytes): a
// caller that allocates from a pool (torch's caching allocator) hands it to
// taoamd_json_pred_open and spares the call its hipMalloc / hipFree -- in a
// process whose allocator holds most of the device those take 0.1 s and more.
extern "C" size_t taoamd_json_pred_workspace(size_t file_bytes)
{
    const size_t n_blk = (file_bytes + JS_BLK - 1) / JS_BLK;
    return n_blk * JS_BLK + 256 + ((n_blk * 6 + 8) * 4 + 256) + (((size_t)2 << 16) + 256) +
           (file_bytes / 32 + 1) * 8 + 1024;
}

extern "C" void *taoamd_json_pred_open(const char *path, void *work, size_t work_bytes,
                                       int32_t *status, char *err, size_t errlen, void *stream)
{
    int32_t dummy;
    if (!status) status = &dummy;
    *status = TAOAMD_OK;
    std::unique_ptr<JsHandle> h(new JsHandle);
    h->s = (hipStream_t)stream;
    if (work && work_bytes) {
        const uintptr_t a0 = ((uintptr_t)work + 255) & ~(uintptr_t)255;
        if (a0 - (uintptr_t)work < work_bytes) {
            h->work = (uint8_t *)a0;
            h->work_left = work_bytes - (a0 - (uintptr_t)work);
        }
    }
    const bool timing = getenv("TAOAMD_INGEST_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(
                        std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    {
        // no GPU in this process: the host reader's file
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) {
            (void)hipGetLastError();
            *status = TAOAMD_JSON_FALLBACK;
            return nullptr;
        }
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        js_msg(err, errlen, std::string("cannot open ") + path);
        *status = TAOAMD_ERR_ARG;
        return nullptr;
    }
    struct stat st;
    fstat(fd, &st);
    h->len = (size_t)st.st_size;
    if (h->len == 0) {
        close(fd);
        *status = TAOAMD_JSON_FALLBACK;
        return nullptr;
    }
    h->map = (const char *)mmap(nullptr, h->len, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (h->map == MAP_FAILED) {
        h->map = nullptr;
        js_msg(err, errlen, "mmap failed");
        *status = TAOAMD_ERR_ARG;
        return nullptr;
    }
    const int64_t n_blk64 = ((int64_t)h->len + JS_BLK - 1) / JS_BLK;
    if (n_blk64 >= INT32_MAX) {
        *status = TAOAMD_JSON_FALLBACK;
        return nullptr;
    }
    const int32_t n_blk = (int32_t)n_blk64;
    const size_t padded = (size_t)n_blk * JS_BLK;
    h->d_text = h->carve(padded + 64);
    h->own_text = h->d_text == nullptr;
    if (h->own_text) JS_ALLOC(hipMalloc(&h->d_text, padded + 64));
    // the text travels in slices, by several threads (the file's pages are
    // mapped on first touch: the copying threads' own faults)
    {
        madvise((void *)h->map, h->len, MADV_WILLNEED);
        Toucher ahead;
        ahead.read(h->map, h->len, 4);
        const hipError_t ce = parallel_copy({Piece{(char *)h->d_text, h->map, h->len}},
                                            hipMemcpyHostToDevice, copy_threads());
        ahead.join();
        if (ce != hipSuccess) {
            taoamd::set_error(ce, "hipMemcpyAsync(text)");
            js_msg(err, errlen, "HIP: copy of the text");
            *status = TAOAMD_ERR_HIP;
            return nullptr;
        }
    }
    JS_HIP(hipMemsetAsync(h->d_text + h->len, ' ', padded + 64 - h->len, h->s));
    const double t1 = now();
    // per-block tables: parity, depth change, object count, their prefix sums
    int32_t *tab = (int32_t *)h->carve(((size_t)n_blk * 6 + 8) * sizeof(int32_t));
    struct Free {
        void *p;
        ~Free() { if (p) (void)hipFree(p); }
    } free_tab{nullptr};
    if (!tab) {
        JS_ALLOC(hipMalloc(&tab, ((size_t)n_blk * 6 + 8) * sizeof(int32_t)));
        free_tab.p = tab;
    }
    h->d_flag = (int64_t *)h->carve((size_t)2 << 16);
    int32_t *par = tab, *delta = par + n_blk, *count = delta + n_blk;
    int32_t *par_ex = count + n_blk, *depth_ex = par_ex + n_blk + 1,
            *count_ex = depth_ex + n_blk + 1;
    uint32_t *bad = (uint32_t *)(count_ex + n_blk + 1), *open0 = bad + 1;
    JS_HIP(hipMemsetAsync(bad, 0, 8, h->s));
    JsArgs a{};
    a.text = (const uint32_t *)h->d_text;
    a.len = (int64_t)h->len;
    a.n_blk = n_blk;
    a.blk_par = par;
    a.blk_delta = delta;
    a.blk_count = count;
    a.par_ex = par_ex;
    a.depth_ex = depth_ex;
    a.count_ex = count_ex;
    a.bad = bad;
    a.n_open0 = open0;
    js_quotes_kernel<<<n_blk, JS_T, 0, h->s>>>(a);
    js_scan_kernel<<<1, 1024, 0, h->s>>>(par, par_ex, n_blk);
    js_walk_kernel<0><<<n_blk, JS_T, 0, h->s>>>(a);
    js_scan_kernel<<<1, 1024, 0, h->s>>>(delta, depth_ex, n_blk);
    js_walk_kernel<1><<<n_blk, JS_T, 0, h->s>>>(a);
    js_scan_kernel<<<1, 1024, 0, h->s>>>(count, count_ex, n_blk);
    int32_t res[4] = {0, 0, 0, 0};      // objects, final depth, anomaly bits, '[' at depth 0
    JS_HIP(hipMemcpyAsync(&res[0], count_ex + n_blk, 4, hipMemcpyDeviceToHost, h->s));
    JS_HIP(hipMemcpyAsync(&res[1], depth_ex + n_blk, 4, hipMemcpyDeviceToHost, h->s));
    JS_HIP(hipMemcpyAsync(&res[2], bad, 8, hipMemcpyDeviceToHost, h->s));
    JS_HIP(hipStreamSynchronize(h->s));
    JS_HIP(hipGetLastError());
    if (res[2] != 0 || res[3] != 1 || res[1] != 0 || res[0] < 0) {
        *status = TAOAMD_JSON_FALLBACK;
        return nullptr;
    }
    h->n = res[0];
    if (h->n > 0) {
        h->d_starts = (int64_t *)h->carve((size_t)h->n * 8);
        h->own_starts = h->d_starts == nullptr;
        if (h->own_starts) JS_ALLOC(hipMalloc(&h->d_starts, (size_t)h->n * 8));
        a.starts = h->d_starts;
        js_walk_kernel<2><<<n_blk, JS_T, 0, h->s>>>(a);
        JS_HIP(hipGetLastError());
    }
    if (timing) {
        (void)hipStreamSynchronize(h->s);
        fprintf(stderr, "taoamd ingest (device): %s: text to HBM %.3f s, %lld objects found %.3f s\n",
                path, t1 - t0, (long long)h->n, now() - t1);
    }
    return h.release();
}

extern "C" int64_t taoamd_json_pred_count(void *h) { return h ? ((JsHandle *)h)->n : 0; }

// The objects converted into the caller's DEVICE arrays of
// taoamd_json_pred_count() rows (bbox: 4 doubles a row); flag[0 ..
// min(*n_flagged, flag_cap)) = the objects left to the host reader (their rows
// are not written), flag_at[] their byte offsets in the file (HOST arrays).
// Returns when the kernel is done.
extern "C" int taoamd_json_pred_convert(void *handle, int64_t *image_id, int64_t *category_id,
                                        double *bbox, double *score, int64_t *track_id,
                                        int64_t *video_id, int64_t *flag, int64_t *flag_at,
                                        int32_t flag_cap, int32_t *n_flagged)
{
    JsHandle *h = (JsHandle *)handle;
    if (!h || !n_flagged || flag_cap < 0 || (flag_cap > 0 && (!flag || !flag_at)))
        return TAOAMD_ERR_ARG;
    *n_flagged = 0;
    const int64_t n = h->n;
    if (n == 0) return TAOAMD_OK;
    if (!image_id || !category_id || !bbox || !score || !track_id || !video_id)
        return TAOAMD_ERR_ARG;
    const bool timing = getenv("TAOAMD_INGEST_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(
                        std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    int64_t *d_flag = (size_t)(2 * flag_cap + 1) * 8 + 64 <= ((size_t)2 << 16) ? h->d_flag : nullptr;
    struct Free {
        void *p;
        ~Free() { if (p) (void)hipFree(p); }
    } free_flag{nullptr};
    if (!d_flag) {
        TAO_HIP(hipMalloc(&d_flag, (size_t)(2 * flag_cap + 1) * 8 + 64));
        free_flag.p = d_flag;
    }
    JsParseArgs a{};
    a.text = h->d_text;
    a.len = (int64_t)h->len;
    a.n = n;
    a.starts = h->d_starts;
    a.c.bbox = bbox;
    a.c.score = score;
    a.c.image_id = image_id;
    a.c.category_id = category_id;
    a.c.track_id = track_id;
    a.c.video_id = video_id;
    a.flag = d_flag;
    a.flag_at = a.flag + flag_cap;
    a.n_flag = (int32_t *)(a.flag_at + flag_cap);
    a.flag_cap = flag_cap;
    TAO_HIP(hipMemsetAsync(a.n_flag, 0, 8, h->s));
    js_parse_kernel<<<(unsigned)((n + 255) / 256), 256, 0, h->s>>>(a);
    TAO_LAUNCH_CHECK();
    int32_t nf = 0;
    TAO_HIP(hipMemcpyAsync(&nf, a.n_flag, 4, hipMemcpyDeviceToHost, h->s));
    TAO_HIP(hipStreamSynchronize(h->s));
    *n_flagged = nf;
    const int32_t k = nf < flag_cap ? nf : flag_cap;
    if (k > 0) {
        TAO_HIP(hipMemcpy(flag, a.flag, (size_t)k * 8, hipMemcpyDeviceToHost));
        TAO_HIP(hipMemcpy(flag_at, a.flag_at, (size_t)k * 8, hipMemcpyDeviceToHost));
    }
    if (timing)
        fprintf(stderr, "taoamd ingest (device): %lld objects converted %.3f s, %d left to the "
                        "host reader\n", (long long)n, now() - t0, (int)nf);
    return TAOAMD_OK;
}

// The same into HOST arrays (device columns of the call's own, copied out:
// the destination's fresh pages are touched by helper threads beside the
// kernel).
extern "C" int taoamd_json_pred_read(void *handle, int64_t *image_id, int64_t *category_id,
                                     double *bbox, double *score, int64_t *track_id,
                                     int64_t *video_id, int64_t *flag, int64_t *flag_at,
                                     int32_t flag_cap, int32_t *n_flagged)
{
    JsHandle *h = (JsHandle *)handle;
    if (!h || !n_flagged) return TAOAMD_ERR_ARG;
    *n_flagged = 0;
    const int64_t n = h->n;
    if (n == 0) return TAOAMD_OK;
    if (!image_id || !category_id || !bbox || !score || !track_id || !video_id)
        return TAOAMD_ERR_ARG;
    const size_t col = (size_t)n * 8;
    if (h->d_cols) return TAOAMD_ERR_ARG;               // (one read per handle)
    TAO_HIP(hipMalloc(&h->d_cols, 9 * col + 64));
    uint8_t *buf = h->d_cols;
    double *d_bbox = (double *)buf;                    // (32-byte rows first: aligned)
    double *d_score = (double *)(buf + 4 * col);
    int64_t *d_img = (int64_t *)(buf + 5 * col), *d_cat = (int64_t *)(buf + 6 * col);
    int64_t *d_trk = (int64_t *)(buf + 7 * col), *d_vid = (int64_t *)(buf + 8 * col);
    Toucher ahead;                 // (beside the kernel)
    ahead.write((char *)bbox, 4 * col, 2);
    for (void *p : {(void *)score, (void *)image_id, (void *)category_id, (void *)track_id,
                    (void *)video_id})
        ahead.write((char *)p, col, 1);
    const int rc = taoamd_json_pred_convert(handle, d_img, d_cat, d_bbox, d_score, d_trk, d_vid,
                                            flag, flag_at, flag_cap, n_flagged);
    ahead.join();
    if (rc != TAOAMD_OK) return rc;
    TAO_HIP(parallel_copy({Piece{(char *)bbox, (const char *)d_bbox, 4 * col},
                           Piece{(char *)score, (const char *)d_score, col},
                           Piece{(char *)image_id, (const char *)d_img, col},
                           Piece{(char *)category_id, (const char *)d_cat, col},
                           Piece{(char *)track_id, (const char *)d_trk, col},
                           Piece{(char *)video_id, (const char *)d_vid, col}},
                          hipMemcpyDeviceToHost, copy_threads()));
    return TAOAMD_OK;
}

extern "C" void taoamd_json_pred_close(void *h)
{
    const auto t0 = std::chrono::steady_clock::now();
    delete (JsHandle *)h;
    if (getenv("TAOAMD_INGEST_TIMING"))
        fprintf(stderr, "taoamd ingest (device): released %.3f s\n",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
}
