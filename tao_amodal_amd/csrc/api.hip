// Status strings, error capture and the threshold tables of the C ABI.
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"

namespace taoamd {

static thread_local char g_err[512] = "";

void set_error(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

// np.linspace(start, stop, num): arange(num) * step + start with
// step = (stop - start) / (num - 1), last sample forced to stop -- the exact
// bit patterns of reference lvis_amodal/eval.py:560-565.
static void fill(double *out, double start, double stop, int num)
{
    const double step = (stop - start) / (double)(num - 1);
    for (int i = 0; i < num; i++) out[i] = (double)i * step + start;
    out[num - 1] = stop;
}

static IouThr default_iou()
{
    IouThr x;
    fill(x.v, 0.5, 0.95, N_THR);
    return x;
}
static RecThr default_rec()
{
    RecThr x;
    fill(x.v, 0.0, 1.0, N_REC);
    return x;
}
static RangeTab default_ranges()
{
    // L/eval.py:567-574 (the sixth, out-of-frame range has no bounds);
    // T/eval.py:735-744
    const RangeTab t = {{0, 0, 0.1, 0.8, 0}, {1.0, 0.1, 0.8, 1.0, 0.8},
                        {0, 0, 1024, 9216, 0}, {1e10, 1024, 9216, 1e10, 1e10},
                        {0, 0, 3, 10}, {1e5, 3, 10, 1e5}};
    return t;
}
// per thread: the drop-in CLI evaluates its two levels on two threads
static thread_local IouThr g_iou = default_iou();
static thread_local RecThr g_rec = default_rec();
static thread_local RangeTab g_rng = default_ranges();

const IouThr &iou_thr() { return g_iou; }
const RecThr &rec_thr() { return g_rec; }
const RangeTab &range_tab() { return g_rng; }

// ---- per-kernel timing: event pairs on the launch streams, reduced on demand
bool g_timing_on = false;
struct TimedLaunch {
    const char *name;
    hipEvent_t a, b;
};
static std::mutex g_timing_mu;
static std::vector<TimedLaunch> g_timed;
static std::vector<hipEvent_t> g_event_pool;
static thread_local TimedLaunch g_open;
static thread_local std::string g_label;
static std::map<std::string, const char *> g_names;      // interned "label:kernel"

static const char *intern(const std::string &nm)
{
    std::lock_guard<std::mutex> lk(g_timing_mu);
    auto it = g_names.find(nm);
    if (it != g_names.end()) return it->second;
    char *c = strdup(nm.c_str());
    g_names[nm] = c;
    return c;
}

static hipEvent_t take_event()
{
    std::lock_guard<std::mutex> lk(g_timing_mu);
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void timing_begin(const char *name, hipStream_t s)
{
    g_open.name = g_label.empty() ? name : intern(g_label + ":" + name);
    g_open.a = take_event();
    g_open.b = take_event();
    (void)hipEventRecord(g_open.a, s);
}

void timing_end(hipStream_t s)
{
    (void)hipEventRecord(g_open.b, s);
    std::lock_guard<std::mutex> lk(g_timing_mu);
    g_timed.push_back(g_open);
}

}  // namespace taoamd

extern "C" int taoamd_kernel_timing_enable(int on)
{
    taoamd::g_timing_on = on != 0;
    return TAOAMD_OK;
}

extern "C" int taoamd_kernel_timing_label(const char *label)
{
    taoamd::g_label = label ? label : "";
    return TAOAMD_OK;
}

extern "C" int taoamd_kernel_timing_collect(char *names, size_t names_bytes,
                                            double *total_ms, int64_t *calls,
                                            int32_t max_kernels,
                                            int32_t *n_kernels)
{
    using namespace taoamd;
    if (!n_kernels || max_kernels < 0) return TAOAMD_ERR_ARG;
    std::vector<TimedLaunch> taken;
    {
        std::lock_guard<std::mutex> lk(g_timing_mu);
        taken.swap(g_timed);
    }
    std::map<std::string, std::pair<double, int64_t>> agg;
    std::vector<std::string> order;
    for (const TimedLaunch &t : taken) {
        TAO_HIP(hipEventSynchronize(t.b));
        float ms = 0.f;
        TAO_HIP(hipEventElapsedTime(&ms, t.a, t.b));
        auto it = agg.find(t.name);
        if (it == agg.end()) {
            order.push_back(t.name);
            agg[t.name] = {ms, 1};
        } else {
            it->second.first += ms;
            it->second.second += 1;
        }
    }
    {
        std::lock_guard<std::mutex> lk(g_timing_mu);
        for (const TimedLaunch &t : taken) {
            g_event_pool.push_back(t.a);
            g_event_pool.push_back(t.b);
        }
    }
    *n_kernels = (int32_t)order.size();
    size_t used = 0;
    int32_t k = 0;
    for (const std::string &nm : order) {
        if (k >= max_kernels) break;
        if (!names || !total_ms || !calls || used + nm.size() + 1 > names_bytes)
            return TAOAMD_ERR_ARG;
        memcpy(names + used, nm.c_str(), nm.size() + 1);   // NUL-separated list
        used += nm.size() + 1;
        total_ms[k] = agg[nm].first;
        calls[k] = agg[nm].second;
        k++;
    }
    return TAOAMD_OK;
}

// ---- events between the caller's streams (taoamd_sort_sampled_notify): plain
// hipEvent_t handles, so that a host that only holds stream handles (ctypes)
// can order its streams around a kernel INSIDE one of the batched calls
extern "C" int taoamd_event_create(void **event)
{
    if (!event) return TAOAMD_ERR_ARG;
    hipEvent_t e = nullptr;
    TAO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = (void *)e;
    return TAOAMD_OK;
}

extern "C" int taoamd_event_destroy(void *event)
{
    if (event) TAO_HIP(hipEventDestroy((hipEvent_t)event));
    return TAOAMD_OK;
}

extern "C" int taoamd_stream_wait_event(void *stream, void *event)
{
    if (!event) return TAOAMD_ERR_ARG;
    TAO_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return TAOAMD_OK;
}

extern "C" const char *taoamd_strerror(int status)
{
    switch (status) {
    case TAOAMD_OK: return "ok";
    case TAOAMD_ERR_HIP: return "HIP runtime error (see taoamd_last_error)";
    case TAOAMD_ERR_ARG: return "bad argument";
    case TAOAMD_ERR_TOO_LARGE: return "a cell exceeds a kernel limit";
    case TAOAMD_ERR_WORKSPACE: return "workspace too small";
    }
    return "unknown status";
}

extern "C" const char *taoamd_last_error(void) { return taoamd::g_err; }

extern "C" int taoamd_version(void) { return 100; }

extern "C" int taoamd_thresholds_host(double *iou_thrs, double *rec_thrs)
{
    if (!iou_thrs || !rec_thrs) return TAOAMD_ERR_ARG;
    const IouThr i = taoamd::default_iou();
    const RecThr r = taoamd::default_rec();
    memcpy(iou_thrs, i.v, sizeof(double) * N_THR);
    memcpy(rec_thrs, r.v, sizeof(double) * N_REC);
    return TAOAMD_OK;
}

extern "C" int taoamd_set_thresholds(const double *iou_thrs, const double *rec_thrs)
{
    using namespace taoamd;
    IouThr i = default_iou();
    RecThr r = default_rec();
    if (iou_thrs) {
        memcpy(i.v, iou_thrs, sizeof(double) * N_THR);
        // the kernels' closed form of the greedy match and the sweeps' tables
        // of recall crossings take the thresholds in ascending order
        for (int k = 0; k < N_THR; k++)
            if (!(i.v[k] == i.v[k]) || (k > 0 && i.v[k] < i.v[k - 1])) return TAOAMD_ERR_ARG;
    }
    if (rec_thrs) {
        memcpy(r.v, rec_thrs, sizeof(double) * N_REC);
        for (int k = 0; k < N_REC; k++)
            if (!(r.v[k] == r.v[k]) || (k > 0 && r.v[k] < r.v[k - 1])) return TAOAMD_ERR_ARG;
    }
    g_iou = i;
    g_rec = r;
    return TAOAMD_OK;
}

extern "C" int taoamd_set_ranges(const double *visibility_rng, const double *area_rng,
                                 const double *time_rng)
{
    using namespace taoamd;
    RangeTab t = default_ranges();
    if (visibility_rng)
        for (int k = 0; k < 5; k++) {
            t.vis_lo[k] = visibility_rng[2 * k];
            t.vis_hi[k] = visibility_rng[2 * k + 1];
        }
    if (area_rng)
        for (int k = 0; k < 5; k++) {
            t.area_lo[k] = area_rng[2 * k];
            t.area_hi[k] = area_rng[2 * k + 1];
        }
    if (time_rng)
        for (int k = 0; k < 4; k++) {
            t.time_lo[k] = time_rng[2 * k];
            t.time_hi[k] = time_rng[2 * k + 1];
        }
    g_rng = t;
    return TAOAMD_OK;
}
