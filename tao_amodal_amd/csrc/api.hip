// Status strings, error capture and the threshold tables of the C ABI.
#include <stdio.h>
#include <string.h>

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

const IouThr &iou_thr()
{
    static IouThr t = [] { IouThr x; fill(x.v, 0.5, 0.95, N_THR); return x; }();
    return t;
}

const RecThr &rec_thr()
{
    static RecThr t = [] { RecThr x; fill(x.v, 0.0, 1.0, N_REC); return x; }();
    return t;
}

}  // namespace taoamd

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
    memcpy(iou_thrs, taoamd::iou_thr().v, sizeof(double) * N_THR);
    memcpy(rec_thrs, taoamd::rec_thr().v, sizeof(double) * N_REC);
    return TAOAMD_OK;
}
