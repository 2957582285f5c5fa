// Cell-table build on the device, detection side (gfx950).
//
// The reference answers every (image, category) / (video, category) query
// through Python dicts (lvis_amodal/lvis.py:34-61,90-96, results.py:20-84,
// eval.py:59-110; tao_amodal/tao.py, results.py).  The ground-truth side of
// that is small (10^5 rows) and full of id corner cases, and stays on the
// host (flatten.py); the prediction side is the bulk -- 3 M boxes at Config 2,
// 30 M at full-validation scale -- and is a handful of streaming passes plus
// two stable radix sorts (sort.hip):
//
//   fl_map      raw ids -> image / category indices (binary search in the
//               sorted unique ids), box area, boxes per image
//                                                       L/results.py:25-52
//   fl_starts   exclusive scan of the per-image counts, their maximum
//   fl_rank     an image with more than max_dets boxes keeps its best max_dets
//               by score, ties to the earlier box (from the stable sort by
//               (image, -score))                        L/results.py:39-40,73-84
//   fl_filter   known category, 0 < area < inf (L/lvis.py:90-96), federated
//               filter (L/eval.py:99-103), dt_ig flags; key = category * U +
//               image for kept boxes, INT32_MAX for dropped ones
//   [stable sort by (key, -score): cells category-major, score order inside,
//    ties in file order -- L/eval.py:175]
//   fl_gather   rows of the kept boxes in final order
//   fl_runs     runs of equal keys (= cells that hold detections): three-phase
//               scan of the run heads
//
// HBM-bound integer work; every pass streams its inputs once.
#include "common.hpp"

using namespace taoamd;

#define FL_THREADS 256
#define FL_TILE 2048

// position of x in the sorted unique ids, -1 when absent
__device__ __forceinline__ int32_t fl_index_of(const int64_t *__restrict__ ids,
                                               int64_t n, int64_t x)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ids[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo < n && ids[lo] == x ? (int32_t)lo : -1;
}

__global__ void fl_map_kernel(int64_t n, const int64_t *__restrict__ image_id,
                              const int64_t *__restrict__ category_id,
                              const double4 *__restrict__ bbox,
                              const double *__restrict__ area_in,
                              int64_t n_img, const int64_t *__restrict__ img_ids,
                              int64_t n_cat, const int64_t *__restrict__ cat_ids,
                              int32_t *__restrict__ img, int32_t *__restrict__ cat,
                              double *__restrict__ area,
                              int32_t *__restrict__ img_count,
                              int32_t *__restrict__ status)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t im = fl_index_of(img_ids, n_img, image_id[i]);
    const int32_t ct = fl_index_of(cat_ids, n_cat, category_id[i]);
    img[i] = im;
    cat[i] = ct;
    if (area_in) {
        area[i] = area_in[i];
    } else {
        const double4 b = bbox[i];
        area[i] = b.z * b.w;                 // L/results.py:51
    }
    if (im < 0) atomicAdd(&status[0], 1);
    else atomicAdd(&img_count[im], 1);
}

// one workgroup: exclusive scan of count[0..n) and its maximum
__global__ __launch_bounds__(1024) void fl_starts_kernel(
    int64_t n, const int32_t *__restrict__ count, int32_t *__restrict__ start,
    int32_t *__restrict__ status)
{
    __shared__ int32_t part[1024];
    __shared__ int32_t pmax[1024];
    const int64_t per = (n + 1023) / 1024;
    const int64_t lo = threadIdx.x * per, hi = min(lo + per, n);
    int32_t s = 0, m = 0;
    for (int64_t i = lo; i < hi; i++) {
        s += count[i];
        m = max(m, count[i]);
    }
    part[threadIdx.x] = s;
    pmax[threadIdx.x] = m;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int32_t v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
        const int32_t w = threadIdx.x >= (unsigned)off ? pmax[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        pmax[threadIdx.x] = max(pmax[threadIdx.x], w);
        __syncthreads();
    }
    int32_t run = part[threadIdx.x] - s;
    for (int64_t i = lo; i < hi; i++) {
        const int32_t c = count[i];
        start[i] = run;
        run += c;
    }
    if (threadIdx.x == 1023) {
        status[1] = pmax[1023];
        if (n >= 0) start[n] = part[1023];
    }
}

__global__ void fl_rank_kernel(int64_t n, const int32_t *__restrict__ order,
                               const int32_t *__restrict__ img,
                               const int32_t *__restrict__ img_start,
                               int32_t max_dets, uint8_t *__restrict__ dropped)
{
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t d = order[p];
    const int32_t im = img[d];
    // (boxes of unknown images sort first: key -1 is rejected before this)
    dropped[d] = im >= 0 && p - img_start[im] >= max_dets;
}

__device__ __forceinline__ bool fl_in_sorted(const int32_t *__restrict__ keys,
                                             int64_t n, int32_t x)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo < n && keys[lo] == x;
}

__device__ __forceinline__ bool fl_in_list(const int64_t *__restrict__ off,
                                           const int64_t *__restrict__ val,
                                           int32_t row, int64_t item)
{
    if (!off || row < 0) return false;
    for (int64_t k = off[row]; k < off[row + 1]; k++)
        if (val[k] == item) return true;
    return false;
}

__global__ void fl_filter_kernel(
    int64_t n, const int32_t *__restrict__ unit, const int32_t *__restrict__ cat,
    const double *__restrict__ area, const int64_t *__restrict__ category_id,
    const uint8_t *__restrict__ dropped, int32_t n_unit, int64_t n_gkeys,
    const int32_t *__restrict__ gkeys, const int32_t *__restrict__ unit_row,
    const int64_t *__restrict__ neg_off, const int64_t *__restrict__ neg_val,
    const int64_t *__restrict__ nel_off, const int64_t *__restrict__ nel_val,
    int32_t area_flag, int32_t *__restrict__ key, uint8_t *__restrict__ flags,
    int32_t *__restrict__ n_keep)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    bool keep = false;
    if (i < n) {
        const int32_t u = unit[i], c = cat[i];
        const double a = area[i];
        keep = u >= 0 && c >= 0 && !(dropped && dropped[i]) && a > 0 && a < INFINITY;
        int32_t k = INT32_MAX;
        uint8_t f = 0;
        if (keep) {
            k = c * n_unit + u;
            const int32_t row = unit_row[u];
            keep = fl_in_sorted(gkeys, n_gkeys, k) ||
                   fl_in_list(neg_off, neg_val, row, category_id[i]);
            if (keep) {
                const bool nel = fl_in_list(nel_off, nel_val, row, category_id[i]);
                // L/eval.py:281-288: outside [0, 1e5**2] or not exhaustively labelled
                f = (nel || (area_flag && (a < 0 || a > 1e10))) ? 1 : 0;
            } else {
                k = INT32_MAX;
            }
        }
        key[i] = k;
        flags[i] = f;
    }
    const uint64_t b = __ballot(keep);
    if (lane_id() == 0 && b) atomicAdd(n_keep, (int32_t)__popcll(b));
}

__global__ void fl_gather_kernel(int64_t n_keep, const int32_t *__restrict__ order,
                                 const double *__restrict__ score,
                                 const uint8_t *__restrict__ flags,
                                 const int32_t *__restrict__ key,
                                 const double4 *__restrict__ bbox, int32_t n_unit,
                                 int32_t *__restrict__ dt_row,
                                 double *__restrict__ dt_score,
                                 uint8_t *__restrict__ dt_flags,
                                 int32_t *__restrict__ dt_key,
                                 int32_t *__restrict__ dt_cat,
                                 double4 *__restrict__ dt_box)
{
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= n_keep) return;
    const int32_t d = order[p];
    const int32_t k = key[d];
    dt_row[p] = d;
    dt_score[p] = score[d];
    dt_flags[p] = flags[d];
    dt_key[p] = k;
    dt_cat[p] = k / n_unit;
    if (dt_box) dt_box[p] = bbox[d];
}

// ---- runs of equal keys in a sorted array: heads -> three-phase scan
__global__ __launch_bounds__(FL_THREADS) void fl_heads_kernel(
    int64_t n, const int32_t *__restrict__ key, int32_t *__restrict__ block_sum)
{
    __shared__ int32_t part[FL_THREADS / WAVE];
    const int64_t base = (int64_t)blockIdx.x * FL_TILE;
    int32_t c = 0;
    for (int k = 0; k < FL_TILE / FL_THREADS; k++) {
        const int64_t i = base + k * FL_THREADS + threadIdx.x;
        if (i < n && (i == 0 || key[i] != key[i - 1])) c++;
    }
    for (int s = WAVE / 2; s > 0; s >>= 1) c += __shfl_down(c, s, WAVE);
    if (lane_id() == 0) part[threadIdx.x / WAVE] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t t = 0;
        for (int w = 0; w < FL_THREADS / WAVE; w++) t += part[w];
        block_sum[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(FL_THREADS) void fl_runs_kernel(
    int64_t n, const int32_t *__restrict__ key,
    const int32_t *__restrict__ block_start, int32_t *__restrict__ run_id,
    int32_t *__restrict__ run_key, int32_t *__restrict__ run_start)
{
    __shared__ int32_t wave_sum[FL_THREADS / WAVE];
    __shared__ int32_t carry;
    const int64_t base = (int64_t)blockIdx.x * FL_TILE;
    if (threadIdx.x == 0) carry = block_start[blockIdx.x];
    __syncthreads();
    for (int k = 0; k < FL_TILE / FL_THREADS; k++) {
        const int64_t i = base + k * FL_THREADS + threadIdx.x;
        const bool head = i < n && (i == 0 || key[i] != key[i - 1]);
        const uint64_t b = __ballot(head);
        const int lane = lane_id(), wave = threadIdx.x / WAVE;
        if (lane == 0) wave_sum[wave] = (int32_t)__popcll(b);
        __syncthreads();
        int32_t before = carry;
        for (int w = 0; w < wave; w++) before += wave_sum[w];
        // inclusive count of heads up to and including this element
        const int32_t incl = before + (int32_t)__popcll(b & ((2ull << lane) - 1));
        if (i < n) {
            run_id[i] = incl - 1;
            if (head) {
                run_key[incl - 1] = key[i];
                run_start[incl - 1] = (int32_t)i;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int32_t t = 0;
            for (int w = 0; w < FL_THREADS / WAVE; w++) t += wave_sum[w];
            carry += t;
        }
        __syncthreads();
    }
}

__global__ void fl_remap_kernel(int64_t n, const int32_t *__restrict__ id,
                                const int32_t *__restrict__ map,
                                int32_t *__restrict__ out)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = map[id[i]];
}

static unsigned fl_blocks(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

extern "C" int taoamd_flat_map(int64_t n, const int64_t *image_id,
                               const int64_t *category_id, const double *bbox,
                               const double *area_in, int64_t n_img,
                               const int64_t *img_ids, int64_t n_cat,
                               const int64_t *cat_ids, int32_t *img, int32_t *cat,
                               double *area, int32_t *img_count,
                               int32_t *img_start, int32_t *status, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0 || n_img < 0 || n > 0x7fffffff) return TAOAMD_ERR_ARG;
    if (!status || !img_count || !img_start) return TAOAMD_ERR_ARG;
    TAO_HIP(hipMemsetAsync(status, 0, 4 * sizeof(int32_t), s));
    TAO_HIP(hipMemsetAsync(img_count, 0, (size_t)(n_img + 1) * sizeof(int32_t), s));
    if (n > 0) {
        if (!image_id || !category_id || (!bbox && !area_in) || !img_ids || !cat_ids ||
            !img || !cat || !area || n_cat < 0)
            return TAOAMD_ERR_ARG;
        TAO_TIMED("fl_map_kernel", s, fl_map_kernel<<<fl_blocks(n, 256), 256, 0, s>>>(
            n, image_id, category_id, (const double4 *)bbox, area_in, n_img, img_ids,
            n_cat, cat_ids, img, cat, area, img_count, status));
    }
    TAO_TIMED("fl_starts_kernel", s, fl_starts_kernel<<<1, 1024, 0, s>>>(
        n_img, img_count, img_start, status));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_rank_drop(int64_t n, const int32_t *order,
                                     const int32_t *img, const int32_t *img_start,
                                     int32_t max_dets, uint8_t *dropped, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0) return TAOAMD_ERR_ARG;
    if (n == 0) return TAOAMD_OK;
    if (!order || !img || !img_start || !dropped) return TAOAMD_ERR_ARG;
    TAO_TIMED("fl_rank_kernel", s, fl_rank_kernel<<<fl_blocks(n, 256), 256, 0, s>>>(
        n, order, img, img_start, max_dets, dropped));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_filter(int64_t n, const int32_t *unit, const int32_t *cat,
                                  const double *area, const int64_t *category_id,
                                  const uint8_t *dropped, int32_t n_unit,
                                  int64_t n_gkeys, const int32_t *gkeys,
                                  const int32_t *unit_row, const int64_t *neg_off,
                                  const int64_t *neg_val, const int64_t *nel_off,
                                  const int64_t *nel_val, int32_t area_flag,
                                  int32_t *key, uint8_t *flags, int32_t *n_keep,
                                  void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0 || n_unit < 0 || n_gkeys < 0 || !n_keep) return TAOAMD_ERR_ARG;
    TAO_HIP(hipMemsetAsync(n_keep, 0, sizeof(int32_t), s));
    if (n == 0) return TAOAMD_OK;
    if (!unit || !cat || !area || !category_id || !unit_row || !key || !flags ||
        (n_gkeys > 0 && !gkeys))
        return TAOAMD_ERR_ARG;
    TAO_TIMED("fl_filter_kernel", s, fl_filter_kernel<<<fl_blocks(n, 256), 256, 0, s>>>(
        n, unit, cat, area, category_id, dropped, n_unit, n_gkeys, gkeys, unit_row,
        neg_off, neg_val, nel_off, nel_val, area_flag, key, flags, n_keep));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_gather(int64_t n_keep, const int32_t *order,
                                  const double *score, const uint8_t *flags,
                                  const int32_t *key, const double *bbox,
                                  int32_t n_unit, int32_t *dt_row, double *dt_score,
                                  uint8_t *dt_flags, int32_t *dt_key, int32_t *dt_cat,
                                  double *dt_box, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n_keep < 0 || n_unit <= 0) return TAOAMD_ERR_ARG;
    if (n_keep == 0) return TAOAMD_OK;
    if (!order || !score || !flags || !key || !dt_row || !dt_score || !dt_flags ||
        !dt_key || !dt_cat || (dt_box && !bbox))
        return TAOAMD_ERR_ARG;
    TAO_TIMED("fl_gather_kernel", s, fl_gather_kernel<<<fl_blocks(n_keep, 256), 256, 0, s>>>(
        n_keep, order, score, flags, key, (const double4 *)bbox, n_unit, dt_row,
        dt_score, dt_flags, dt_key, dt_cat, (double4 *)dt_box));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" size_t taoamd_flat_runs_workspace(int64_t n)
{
    const size_t nb = (size_t)((n < 1 ? 1 : n) + FL_TILE - 1) / FL_TILE;
    return 2 * ((nb + 2) * 4 + 256) + 256;
}

extern "C" int taoamd_flat_runs(int64_t n, const int32_t *sorted_key, int32_t *run_id,
                                int32_t *run_key, int32_t *run_start, int32_t *n_runs,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0 || !n_runs) return TAOAMD_ERR_ARG;
    if (n == 0) {
        TAO_HIP(hipMemsetAsync(n_runs, 0, sizeof(int32_t), s));
        return TAOAMD_OK;
    }
    if (!sorted_key || !run_id || !run_key || !run_start || !workspace)
        return TAOAMD_ERR_ARG;
    if (workspace_bytes < taoamd_flat_runs_workspace(n)) return TAOAMD_ERR_WORKSPACE;
    const int64_t nb = (n + FL_TILE - 1) / FL_TILE;
    unsigned char *w = (unsigned char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    int32_t *block_sum = (int32_t *)w;
    int32_t *block_start = (int32_t *)(w + (((size_t)(nb + 2) * 4 + 255) & ~(size_t)255));
    TAO_TIMED("fl_heads_kernel", s, fl_heads_kernel<<<(unsigned)nb, FL_THREADS, 0, s>>>(
        n, sorted_key, block_sum));
    // exclusive scan of the block sums; the total lands in block_start[nb]
    // (fl_starts_kernel also writes a maximum to status[1]: the spare word
    // behind the sums)
    TAO_TIMED("fl_starts_kernel", s, fl_starts_kernel<<<1, 1024, 0, s>>>(
        nb, block_sum, block_start, block_sum + nb - 1));
    TAO_TIMED("fl_runs_kernel", s, fl_runs_kernel<<<(unsigned)nb, FL_THREADS, 0, s>>>(
        n, sorted_key, block_start, run_id, run_key, run_start));
    TAO_HIP(hipMemcpyAsync(n_runs, block_start + nb, sizeof(int32_t),
                           hipMemcpyDeviceToDevice, s));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_remap(int64_t n, const int32_t *id, const int32_t *map,
                                 int32_t *out, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0) return TAOAMD_ERR_ARG;
    if (n == 0) return TAOAMD_OK;
    if (!id || !map || !out) return TAOAMD_ERR_ARG;
    TAO_TIMED("fl_remap_kernel", s, fl_remap_kernel<<<fl_blocks(n, 256), 256, 0, s>>>(
        n, id, map, out));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}
