// IoU of run-length masks, per (detection, ground truth) pair of every cell:
// the arithmetic mask_utils.iou reaches for iou_type="segm" (reference
// lvis_amodal/eval.py:179-191 -> pycocotools rleIou, maskApi.c:77-96, with
// iscrowd = 0).
//
// The reference walks the two run lists of a pair with two cursors -- a chain
// of ka + kb dependent steps.  The same integers come out of a formulation
// without a chain.  Per mask, once: E[r] = pixels up to the end of run r,
// P[r] = ones up to the end of run r (rle_prefix_kernel, one wavefront per
// mask).  Ones of B before pixel x:  F_B(x) = P_B[r] - (r odd ? E_B[r] - x : 0)
// with r the first run that ends beyond x (binary search).  Then
//     |A & B| = sum over A's runs of ones [s, e) of F_B(e) - F_B(s)
//             = sum_i  (i odd ? +1 : -1) * F_B(E_A[i])      (an even i counts
//               only when a run of ones follows it)
//     |A | B| = |A| + |B| - |A & B|
// so the boundaries of A can be dealt to the lanes of a wavefront, which adds
// the signed terms up; all of it is unsigned 32-bit arithmetic modulo 2^32
// like the reference's counters.  A lane owns 8 CONSECUTIVE boundaries: one
// binary search places the first, the others follow B's runs forward (both
// lists are sorted -- a merge), crossing a long stretch by bisection.
// One workgroup per cell, one wavefront per detection at a time; the 64 lanes
// first test 64 ground truths at once (tight boxes, frame sizes: 0 / -1 are
// stored right away) and only the pairs left are walked; the (E, P) pairs of
// the cell's ground truths wait in LDS when they fit (6144 runs), else they
// are read through the caches.  Integer adds / compares on data read once: no
// use for the matrix cores; measured bound by instruction issue and LDS
// latency, not by HBM.
#include "common.hpp"

using namespace taoamd;

#define RLE_LDS_RUNS 6144

struct RleArgs {
    const int32_t *cell_dt_off, *cell_gt_off;
    const int64_t *cell_iou_off;
    const int64_t *dt_off, *gt_off;           // CSR of the run lists
    const int32_t *dt_hw, *gt_hw;             // (height, width) per mask
    const double *dt_bb, *gt_bb;              // tight box (x, y, w, h) per mask
    const uint32_t *dt_end;                   // E per run of the detections
    const uint2 *gt_pre;                      // (E, P) per run of the ground truths
    const uint32_t *dt_ones, *gt_ones;        // ones per mask
    double *iou;
};

// one wavefront per mask: inclusive scans of the run lengths and of the
// lengths of the odd-numbered runs (the ones)
// (detections are only ever the "A" of a pair: their P is not stored.  Summing
// their boundaries inside the IoU kernel instead, lazily for the detections
// that have a pair to walk, was measured slower: 0.56 vs 0.54 ms.)
template <bool WITH_P>
__global__ __launch_bounds__(256) void rle_prefix_kernel(int64_t n,
                                                         const int64_t *off,
                                                         const uint32_t *runs,
                                                         void *out, uint32_t *ones)
{
    const int lane = lane_id();
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= n) return;
    const int64_t b = off[m], k = off[m + 1] - b;
    uint32_t carry_e = 0, carry_p = 0;
    for (int64_t base = 0; base < k; base += WAVE) {
        const int64_t i = base + lane;
        const uint32_t c = i < k ? runs[b + i] : 0;
        uint32_t e = c, p = (i & 1) ? c : 0;
#pragma unroll
        for (int s = 1; s < WAVE; s <<= 1) {
            const uint32_t ue = __shfl_up(e, s, WAVE), up = __shfl_up(p, s, WAVE);
            if (lane >= s) { e += ue; p += up; }
        }
        e += carry_e;
        p += carry_p;
        if (i < k) {
            if (WITH_P) ((uint2 *)out)[b + i] = make_uint2(e, p);
            else ((uint32_t *)out)[b + i] = e;
        }
        carry_e = __shfl(e, WAVE - 1, WAVE);
        carry_p = __shfl(p, WAVE - 1, WAVE);
    }
    if (lane == 0) ones[m] = carry_p;
}

#ifndef RLE_THREADS
#define RLE_THREADS 512         // 8 wavefronts (256: 0.77 ms, 1024: 0.64 ms, 512: 0.57 ms)
#endif
#define RLE_RPT 8               // run boundaries of A a lane keeps in registers (even)
#define RLE_GTILE 64            // ground truths looked at together (one per lane)

// One workgroup = one cell, one wavefront = one detection at a time.  For a
// detection the 64 lanes first test 64 ground truths at once (tight boxes,
// frame sizes) and store the 0 / -1 results; the few pairs left are walked:
// every lane holds up to RLE_RPT run boundaries of the detection in registers
// (loaded once, used for all its pairs): one binary search places the first,
// the others follow B's runs forward.
template <bool IN_LDS>
__device__ __forceinline__ void rle_cell(const RleArgs &a, int32_t d0, int32_t D,
                                         int32_t g0, int32_t G,
                                         const uint2 *__restrict__ gt_tab,
                                         int64_t gt_tab_base)
{
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    double *__restrict__ out = a.iou + a.cell_iou_off[blockIdx.x];
    for (int32_t dd = wave; dd < D; dd += RLE_THREADS / WAVE) {
        const int32_t d = d0 + dd;
        const double *db = a.dt_bb + 4 * (int64_t)d;
        const double dx = db[0], dy = db[1], dw = db[2], dh = db[3];
        const int32_t hd = a.dt_hw[2 * d], wd = a.dt_hw[2 * d + 1];
        const int64_t ab = a.dt_off[d];
        const uint32_t ka = (uint32_t)(a.dt_off[d + 1] - ab);
        const uint32_t ones_a = a.dt_ones[d];
        // the detection's run boundaries, RLE_RPT * 64 at a time; a list that
        // fits one piece (the usual case) is loaded once for all its pairs.
        // A lane owns RLE_RPT CONSECUTIVE boundaries i = piece base + lane*RPT
        // + q (RLE_RPT is even: the parity of i is q's).  An even one (start
        // of a run of ones) only counts when that run exists.
        const uint32_t n_pieces = (ka + RLE_RPT * WAVE - 1) / (RLE_RPT * WAVE);
        uint32_t xq[RLE_RPT];
        bool use[RLE_RPT];
        auto load_piece = [&](uint32_t piece) {
#pragma unroll
            for (int q = 0; q < RLE_RPT; q++) {
                const uint32_t i = piece * (RLE_RPT * WAVE) + (uint32_t)lane * RLE_RPT + q;
                use[q] = i < ka && ((i & 1) || i + 1 < ka);
                // past the end: 0, a boundary that makes the walk below stand still
                xq[q] = i < ka ? a.dt_end[ab + i] : 0u;
            }
        };
        load_piece(0);
        for (int32_t gbase = 0; gbase < G; gbase += RLE_GTILE) {
            // lane = one ground truth of the tile
            const int32_t gl = gbase + lane;
            bool walk = false;
            if (gl < G) {
                const int32_t g = g0 + gl;
                const double *gb = a.gt_bb + 4 * (int64_t)g;
                const double w = fmin(dw + dx, gb[2] + gb[0]) - fmax(dx, gb[0]);
                const double h = fmin(dh + dy, gb[3] + gb[1]) - fmax(dy, gb[1]);
                if (!(w > 0) || !(h > 0))
                    out[(int64_t)dd * G + gl] = 0.0;
                else if (hd != a.gt_hw[2 * g] || wd != a.gt_hw[2 * g + 1])
                    out[(int64_t)dd * G + gl] = -1.0;    // other frame size
                else
                    walk = true;
            }
            for (uint64_t live = __ballot(walk); live != 0; live &= live - 1) {
                const int32_t gg = gbase + __builtin_ctzll(live);
                const int32_t g = g0 + gg;
                const int64_t bb = a.gt_off[g];
                const uint32_t kb = (uint32_t)(a.gt_off[g + 1] - bb);
                const uint2 *__restrict__ tab = gt_tab + (bb - gt_tab_base);
                const uint32_t ones_b = a.gt_ones[g];
                uint32_t acc = 0;
                for (uint32_t piece = 0; piece < n_pieces; piece++) {
                    if (n_pieces > 1) load_piece(piece);
                    // one binary search per lane, for its first boundary
                    // (branch-free: ceil(log2(kb + 1)) halvings) ...
                    uint32_t lo = 0, hi = kb;
                    for (uint32_t span = kb; span != 0; span >>= 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        const uint32_t e = tab[min(mid, kb - 1)].x;
                        const bool open = lo < hi, right = e <= xq[0];
                        lo = (open && right) ? mid + 1 : lo;
                        hi = (open && !right) ? mid : hi;
                    }
                    // ... then B's runs are followed while the lane's own
                    // boundaries rise: both lists are sorted, so all of it is
                    // one merge of ~RLE_RPT * (1 + kb / ka) steps per lane.
                    // (Searching every boundary on its own costs ~10 rounds
                    // x 12 VALU instructions each: measured 4x slower, the
                    // kernel is bound by instruction issue, not by the LDS.)
                    uint32_t r = lo;
                    uint2 cur = tab[min(r, kb - 1)];
#pragma unroll
                    for (int q = 0; q < RLE_RPT; q++) {
                        if (!use[q]) continue;      // (also the trailing run of zeros)
                        const uint32_t x = xq[q];
                        // a few steps usually do; a long stretch of B inside
                        // one run of A (A in two distant parts, A's last run)
                        // is crossed by bisection instead
                        for (int step = 0; step < 4 && r < kb && cur.x <= x; step++) {
                            r++;
                            cur = tab[min(r, kb - 1)];
                        }
                        if (r < kb && cur.x <= x) {
                            uint32_t l2 = r + 1, h2 = kb;
                            while (l2 < h2) {
                                const uint32_t mid = (l2 + h2) >> 1;
                                if (tab[mid].x <= x) l2 = mid + 1; else h2 = mid;
                            }
                            r = l2;
                            cur = tab[min(r, kb - 1)];
                        }
                        const uint32_t part = (r & 1) ? cur.x - x : 0u;
                        const uint32_t f = r < kb ? cur.y - part : ones_b;
                        acc += (q & 1) ? f : 0u - f;
                    }
                }
#pragma unroll
                for (int s = WAVE / 2; s > 0; s >>= 1) acc += __shfl_xor(acc, s, WAVE);
                const uint32_t inter = acc;
                const uint32_t uni = inter == 0 ? 1u : ones_a + ones_b - inter;
                if (lane == 0) out[(int64_t)dd * G + gg] = (double)inter / (double)uni;
            }
        }
    }
}

__global__ __launch_bounds__(RLE_THREADS) void rle_iou_kernel(RleArgs a)
{
    __shared__ uint2 s_tab[RLE_LDS_RUNS];
    const int64_t c = blockIdx.x;
    const int32_t d0 = a.cell_dt_off[c], g0 = a.cell_gt_off[c];
    const int32_t D = a.cell_dt_off[c + 1] - d0, G = a.cell_gt_off[c + 1] - g0;
    if (D == 0 || G == 0) return;
    const int64_t r0 = a.gt_off[g0], R = a.gt_off[g0 + G] - r0;
    if (R <= RLE_LDS_RUNS) {                 // workgroup-uniform
        for (int64_t i = threadIdx.x; i < R; i += RLE_THREADS) s_tab[i] = a.gt_pre[r0 + i];
        __syncthreads();
        rle_cell<true>(a, d0, D, g0, G, s_tab, r0);
    } else {
        rle_cell<false>(a, d0, D, g0, G, a.gt_pre, 0);
    }
}

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t taoamd_rle_iou_workspace(int64_t n_dt, int64_t dt_runs,
                                           int64_t n_gt, int64_t gt_runs)
{
    return up256((size_t)dt_runs * 4) + up256((size_t)gt_runs * 8) +
           up256((size_t)n_dt * 4) + up256((size_t)n_gt * 4) + 256;
}

extern "C" int taoamd_rle_iou(int64_t n_cells, const int32_t *cell_dt_off,
                              const int32_t *cell_gt_off,
                              const int64_t *cell_iou_off, int64_t n_dt,
                              int64_t dt_total, const int64_t *dt_off,
                              const uint32_t *dt_runs, const int32_t *dt_hw,
                              const double *dt_bb, int64_t n_gt, int64_t gt_total,
                              const int64_t *gt_off, const uint32_t *gt_runs,
                              const int32_t *gt_hw, const double *gt_bb,
                              double *iou, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    if (n_cells < 0 || n_dt < 0 || n_gt < 0 || dt_total < 0 || gt_total < 0)
        return TAOAMD_ERR_ARG;
    if (n_cells == 0 || n_dt == 0 || n_gt == 0) return TAOAMD_OK;
    if (!cell_dt_off || !cell_gt_off || !cell_iou_off || !dt_off || !dt_runs ||
        !dt_hw || !dt_bb || !gt_off || !gt_runs || !gt_hw || !gt_bb || !iou ||
        !workspace)
        return TAOAMD_ERR_ARG;
    if (workspace_bytes < taoamd_rle_iou_workspace(n_dt, dt_total, n_gt, gt_total))
        return TAOAMD_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    unsigned char *w = (unsigned char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    RleArgs a;
    a.dt_end = (const uint32_t *)w; w += up256((size_t)dt_total * 4);
    a.gt_pre = (const uint2 *)w; w += up256((size_t)gt_total * 8);
    a.dt_ones = (const uint32_t *)w; w += up256((size_t)n_dt * 4);
    a.gt_ones = (const uint32_t *)w;
    TAO_TIMED("rle_prefix_kernel", s, rle_prefix_kernel<false><<<dim3((unsigned)((n_dt + 3) / 4)), 256, 0, s>>>(
        n_dt, dt_off, dt_runs, (void *)a.dt_end, (uint32_t *)a.dt_ones));
    TAO_LAUNCH_CHECK();
    TAO_TIMED("rle_prefix_kernel", s, rle_prefix_kernel<true><<<dim3((unsigned)((n_gt + 3) / 4)), 256, 0, s>>>(
        n_gt, gt_off, gt_runs, (void *)a.gt_pre, (uint32_t *)a.gt_ones));
    TAO_LAUNCH_CHECK();
    a.cell_dt_off = cell_dt_off; a.cell_gt_off = cell_gt_off;
    a.cell_iou_off = cell_iou_off;
    a.dt_off = dt_off; a.gt_off = gt_off;
    a.dt_hw = dt_hw; a.gt_hw = gt_hw; a.dt_bb = dt_bb; a.gt_bb = gt_bb;
    a.iou = iou;
    TAO_TIMED("rle_iou_kernel", s, rle_iou_kernel<<<dim3((unsigned)n_cells), RLE_THREADS, 0, s>>>(a));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}
