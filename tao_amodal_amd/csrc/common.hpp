// Shared declarations of the gfx950 evaluation kernels.  CDNA4 only: 64-lane
// wavefronts are assumed everywhere (no warp-size abstraction, no CUDA path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tao_amodal_hip.h"

#define WAVE 64
#define N_THR TAOAMD_N_THR
#define N_REC TAOAMD_N_REC

// IoU / recall thresholds travel as kernel arguments (uniform -> SGPRs).
struct IouThr {
    double v[N_THR];
};
struct RecThr {
    double v[N_REC];
};
// Range tables of the two Params classes (L/eval.py:567-574, T/eval.py:735-744):
// by-value kernel arguments like the thresholds.  The COUNTS are the kernels'
// (5 visibility ranges + the out-of-frame one; 5 areas x 4 durations, the last
// area range being the occlusion one); the values are the caller's.
struct RangeTab {
    double vis_lo[5], vis_hi[5];
    double area_lo[5], area_hi[5];
    double time_lo[4], time_hi[4];
};

namespace taoamd {

void set_error(hipError_t e, const char *what);
// the calling thread's evaluation constants: the reference's defaults unless
// taoamd_set_thresholds / taoamd_set_ranges replaced them (params edited by the
// caller of the class API)
const IouThr &iou_thr();
const RecThr &rec_thr();
const RangeTab &range_tab();

#define TAO_HIP(call)                                     \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) {                           \
            taoamd::set_error(e_, #call);                 \
            return TAOAMD_ERR_HIP;                        \
        }                                                 \
    } while (0)

// Optional per-kernel timing (taoamd_kernel_timing_*): a launch wrapped in
// TAO_TIMED is bracketed by two HIP events recorded on ITS stream while timing
// is switched on; switched off (the default) the wrapper costs one load.
extern bool g_timing_on;
void timing_begin(const char *name, hipStream_t s);
void timing_end(hipStream_t s);
struct KernelTimer {
    hipStream_t s;
    bool on;
    KernelTimer(const char *name, hipStream_t s_) : s(s_), on(g_timing_on)
    {
        if (on) timing_begin(name, s);
    }
    ~KernelTimer()
    {
        if (on) timing_end(s);
    }
};
#define TAO_TIMED(name, stream, ...)                      \
    do {                                                  \
        taoamd::KernelTimer tao_timer_(name, stream);     \
        __VA_ARGS__;                                      \
    } while (0)

#define TAO_LAUNCH_CHECK()                                \
    do {                                                  \
        hipError_t e_ = hipGetLastError();                \
        if (e_ != hipSuccess) {                           \
            taoamd::set_error(e_, "kernel launch");       \
            return TAOAMD_ERR_HIP;                        \
        }                                                 \
    } while (0)

// One box pair of bbIou with iscrowd == 0 (reference maskApi.c:109-120).
// Compiled with -ffp-contract=off: da + ga - w*h must NOT become an fma, or
// the last bit of the union differs from the CPU result.
// v_min_f64 / v_max_f64 as such: fmin() / fmax() make the compiler canonicalise
// every operand it cannot prove free of signalling NaNs -- one extra v_max_f64 x, x
// per operand, 6 of the ~45 fp64 instructions of an IoU, in a kernel that is
// bound by VALU issue.  For everything but signalling NaNs (which no parsed or
// computed coordinate is) the instructions return what fmin / fmax return:
// the smaller / larger operand, the other one if one is a NaN.
__device__ __forceinline__ double raw_fmin(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double raw_fmax(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double box_iou(double dx, double dy, double dw,
                                          double dh, double gx, double gy,
                                          double gw, double gh)
{
    double da = dw * dh, ga = gw * gh;
    double w = raw_fmin(dw + dx, gw + gx) - raw_fmax(dx, gx);
    if (w <= 0) return 0.0;
    double h = raw_fmin(dh + dy, gh + gy) - raw_fmax(dy, gy);
    if (h <= 0) return 0.0;
    double i = w * h;
    double u = da + ga - i;
    return i / u;
}

// Smallest TP count c with fl(c / num_gt) >= x, the integer form of
// np.searchsorted(tp / num_gt, x, side="left") (reference
// lvis_amodal/eval.py:386,406-408); ng > 0.
__device__ __forceinline__ int32_t recall_crossing(double x, int32_t ng)
{
    const double dn = (double)ng;
    int32_t c = (int32_t)(x * dn);
    c = c < 0 ? 0 : (c > ng ? ng : c);
    while (c < ng && (double)c / dn < x) c++;
    while (c > 0 && (double)(c - 1) / dn >= x) c--;
    return c;
}

// Precision records.  The sweeps track precision as the integer pair
// (tp, n = tp + fp) packed as tp << 32 | n and leave such records in the
// per-threshold table `val`; whoever writes the reference layout
// (acc_finalize_kernel, ex_unpack_kernel) turns a record into
// tp / (fp + tp + eps), the reference's expression (lvis_amodal/eval.py:384).
// better(a, b): pair a gives a larger value than pair b: fl(tp / (n + eps)) is
// monotone in the rational tp / n, and n + eps == n for n >= 2, so ties go to
// the larger n, which ranks (1, 1) -> 1 / (1 + eps) below (k, k) -> 1.
#define ACC_EPS 2.220446049250313e-16  // np.spacing(1)
#define PR_ZERO 1ull                   // (0, 1): value 0
__device__ __forceinline__ bool pr_better(uint32_t ta, uint32_t na, uint64_t b)
{
    const uint32_t tb = (uint32_t)(b >> 32), nb = (uint32_t)b;
    const uint64_t l = (uint64_t)ta * nb, r = (uint64_t)tb * na;
    return l > r || (l == r && na > nb);
}
__device__ __forceinline__ uint64_t pr_pack(uint32_t t, uint32_t n)
{
    return ((uint64_t)t << 32) | n;
}
__device__ __forceinline__ double pr_value(uint64_t p)
{
    const double t = (double)(uint32_t)(p >> 32), n = (double)(uint32_t)p;
    return t / (n + ACC_EPS);   // fp + tp == n exactly
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// XCD-aware block order.  The dispatcher deals workgroups to the 8 XCDs round
// robin (block b runs on XCD b % 8 -- observed, MI355X_MICROARCH.md; a matter
// of speed only, nothing here depends on it for correctness).  Kernels whose
// consecutive blocks work on one category -- scattered 8..16-byte stores into
// that category's segment of the sorted layout -- want those blocks behind ONE
// L2, so that a line is completed there and leaves for HBM once instead of as
// eight partial lines from eight L2s.  xcd_block() renumbers: XCD x gets the
// x-th contiguous eighth of the logical blocks.  Bijective for any n.
#define N_XCD 8u
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t n)
{
    const uint32_t x = b % N_XCD, q = n / N_XCD, r = n % N_XCD;
    return x * q + (x < r ? x : r) + b / N_XCD;
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

}  // namespace taoamd
