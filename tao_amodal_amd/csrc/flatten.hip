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
                              int32_t *__restrict__ img_first,
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
    if (im < 0) {
        atomicAdd(&status[0], 1);
    } else {
        atomicAdd(&img_count[im], 1);
        if (img_first) atomicMin(&img_first[im], (int32_t)i);
    }
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
template <typename K>
__device__ __forceinline__ bool fl_is_head(const K *__restrict__ key,
                                           const int32_t *__restrict__ order,
                                           int64_t i)
{
    if (i == 0) return true;
    return order ? key[order[i]] != key[order[i - 1]] : key[i] != key[i - 1];
}

template <typename K>
__global__ __launch_bounds__(FL_THREADS) void fl_heads_kernel(
    int64_t n, const K *__restrict__ key, const int32_t *__restrict__ order,
    int32_t *__restrict__ block_sum)
{
    __shared__ int32_t part[FL_THREADS / WAVE];
    const int64_t base = (int64_t)blockIdx.x * FL_TILE;
    int32_t c = 0;
    for (int k = 0; k < FL_TILE / FL_THREADS; k++) {
        const int64_t i = base + k * FL_THREADS + threadIdx.x;
        if (i < n && fl_is_head(key, order, i)) c++;
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

template <typename K>
__global__ __launch_bounds__(FL_THREADS) void fl_runs_kernel(
    int64_t n, const K *__restrict__ key, const int32_t *__restrict__ order,
    const int32_t *__restrict__ block_start, int32_t *__restrict__ run_id,
    K *__restrict__ run_key, int32_t *__restrict__ run_start)
{
    __shared__ int32_t wave_sum[FL_THREADS / WAVE];
    __shared__ int32_t carry;
    const int64_t base = (int64_t)blockIdx.x * FL_TILE;
    if (threadIdx.x == 0) carry = block_start[blockIdx.x];
    __syncthreads();
    for (int k = 0; k < FL_TILE / FL_THREADS; k++) {
        const int64_t i = base + k * FL_THREADS + threadIdx.x;
        const bool head = i < n && fl_is_head(key, order, i);
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
                run_key[incl - 1] = order ? key[order[i]] : key[i];
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
                               int32_t *img_start, int32_t *img_first,
                               int32_t *status, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0 || n_img < 0 || n > 0x7fffffff) return TAOAMD_ERR_ARG;
    if (!status || !img_count || !img_start) return TAOAMD_ERR_ARG;
    TAO_HIP(hipMemsetAsync(status, 0, 4 * sizeof(int32_t), s));
    TAO_HIP(hipMemsetAsync(img_count, 0, (size_t)(n_img + 1) * sizeof(int32_t), s));
    if (img_first)       // 0x7f7f7f7f: larger than any box index in use
        TAO_HIP(hipMemsetAsync(img_first, 0x7f, (size_t)n_img * sizeof(int32_t), s));
    if (n > 0) {
        if (!image_id || !category_id || (!bbox && !area_in) || !img_ids || !cat_ids ||
            !img || !cat || !area || n_cat < 0)
            return TAOAMD_ERR_ARG;
        TAO_TIMED("fl_map_kernel", s, fl_map_kernel<<<fl_blocks(n, 256), 256, 0, s>>>(
            n, image_id, category_id, (const double4 *)bbox, area_in, n_img, img_ids,
            n_cat, cat_ids, img, cat, area, img_count, img_first, status));
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

template <typename K>
static int flat_runs(int64_t n, const K *key, const int32_t *order, int32_t *run_id,
                     K *run_key, int32_t *run_start, int32_t *n_runs,
                     void *workspace, size_t workspace_bytes, hipStream_t s)
{
    if (n < 0 || !n_runs) return TAOAMD_ERR_ARG;
    if (n == 0) {
        TAO_HIP(hipMemsetAsync(n_runs, 0, sizeof(int32_t), s));
        return TAOAMD_OK;
    }
    if (!key || !run_id || !run_key || !run_start || !workspace) return TAOAMD_ERR_ARG;
    if (workspace_bytes < taoamd_flat_runs_workspace(n)) return TAOAMD_ERR_WORKSPACE;
    const int64_t nb = (n + FL_TILE - 1) / FL_TILE;
    unsigned char *w = (unsigned char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    int32_t *block_sum = (int32_t *)w;
    int32_t *block_start = (int32_t *)(w + (((size_t)(nb + 2) * 4 + 255) & ~(size_t)255));
    TAO_TIMED("fl_heads_kernel", s, fl_heads_kernel<K><<<(unsigned)nb, FL_THREADS, 0, s>>>(
        n, key, order, block_sum));
    // exclusive scan of the block sums; the total lands in block_start[nb]
    // (fl_starts_kernel also writes a maximum to status[1]: the spare word
    // behind the sums)
    TAO_TIMED("fl_starts_kernel", s, fl_starts_kernel<<<1, 1024, 0, s>>>(
        nb, block_sum, block_start, block_sum + nb - 1));
    TAO_TIMED("fl_runs_kernel", s, fl_runs_kernel<K><<<(unsigned)nb, FL_THREADS, 0, s>>>(
        n, key, order, block_start, run_id, run_key, run_start));
    TAO_HIP(hipMemcpyAsync(n_runs, block_start + nb, sizeof(int32_t),
                           hipMemcpyDeviceToDevice, s));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_runs(int64_t n, const int32_t *sorted_key, int32_t *run_id,
                                int32_t *run_key, int32_t *run_start, int32_t *n_runs,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    return flat_runs<int32_t>(n, sorted_key, nullptr, run_id, run_key, run_start,
                              n_runs, workspace, workspace_bytes, (hipStream_t)stream);
}

// the same over key[order[i]] (keys read through the sorting permutation)
extern "C" int taoamd_flat_runs_by(int64_t n, const int32_t *key, const int32_t *order,
                                   int32_t *run_id, int32_t *run_key,
                                   int32_t *run_start, int32_t *n_runs,
                                   void *workspace, size_t workspace_bytes,
                                   void *stream)
{
    if (!order) return TAOAMD_ERR_ARG;
    return flat_runs<int32_t>(n, key, order, run_id, run_key, run_start, n_runs,
                              workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int taoamd_flat_runs64_by(int64_t n, const int64_t *key,
                                     const int32_t *order, int32_t *run_id,
                                     int64_t *run_key, int32_t *run_start,
                                     int32_t *n_runs, void *workspace,
                                     size_t workspace_bytes, void *stream)
{
    if (!order) return TAOAMD_ERR_ARG;
    return flat_runs<int64_t>(n, key, order, run_id, run_key, run_start, n_runs,
                              workspace, workspace_bytes, (hipStream_t)stream);
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

// ===========================================================================
// Track level (TaoEval): boxes -> tracks -> cells
// (T/results.py:20-132, T/tao.py:108-254, T/eval.py:196-243)
//
//   fl_ordinal     position of a box inside its image in the post-truncation
//                  list (file order; score order for an image beyond max_dets)
//                  and the top-max_dets cut            T/results.py:56-81,121-132
//   fl_merge_cat   category merge + index lookup       T/tao.py:115-118, results.py:47-50
//   fl_track_of    dense track index of every box from the sort by track id;
//                  a track must stay in one video      T/results.py:111-119
//   fl_keys        sort keys: list order of the kept boxes, visiting order of
//                  the selected ones (CPython set order of the images, built
//                  on the host)                        T/tao.py:224-254
//   fl_track_kept  per track over its kept boxes in list order: one category
//                  (T/results.py:121-132), score = the boxes' common score or
//                  np.mean (numpy's pairwise summation) T/results.py:88-98
//   fl_track_sel   per track over its selected boxes in frame order: length,
//                  left-to-right mean area (T/tao.py:181-187), distinct images,
//                  first appearance
//   fl_track_filter federated filter on the video lists (T/eval.py:214-233)
//   fl_frames      frame lists: one box per image, the last in frame order
//                  (T/eval.py:322-325)
// ===========================================================================
__global__ void fl_ordscore_kernel(int64_t n, const int32_t *__restrict__ img,
                                   const int32_t *__restrict__ img_count,
                                   const double *__restrict__ score,
                                   int32_t max_dets, double *__restrict__ out)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t im = img[i];
    out[i] = (im >= 0 && img_count[im] > max_dets) ? score[i] : 0.0;
}

__global__ void fl_ordinal_kernel(int64_t n, const int32_t *__restrict__ order,
                                  const int32_t *__restrict__ img,
                                  const int32_t *__restrict__ img_start,
                                  int32_t max_dets, int32_t *__restrict__ ordinal,
                                  uint8_t *__restrict__ dropped)
{
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t d = order[p];
    const int32_t im = img[d];
    const int32_t o = im >= 0 ? (int32_t)(p - img_start[im]) : 0;
    ordinal[d] = o;
    dropped[d] = im < 0 || (max_dets >= 0 && o >= max_dets);
}

// boxes of the post-truncation list with a negative corner or an empty side
// (the warning of T/results.py:100-103)
__global__ void fl_count_bad_kernel(int64_t n, const double4 *__restrict__ bbox,
                                    const uint8_t *__restrict__ dropped,
                                    int32_t *__restrict__ count)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    bool bad = false;
    if (i < n && !dropped[i]) {
        const double4 b = bbox[i];
        bad = b.x < 0 || b.y < 0 || b.z <= 0 || b.w <= 0;
    }
    const uint64_t m = __ballot(bad);
    if (lane_id() == 0 && m) atomicAdd(count, (int32_t)__popcll(m));
}

__global__ void fl_merge_cat_kernel(int64_t n, const int64_t *__restrict__ category_id,
                                    int64_t n_merge, const int64_t *__restrict__ msrc,
                                    const int64_t *__restrict__ mdst, int64_t n_cat,
                                    const int64_t *__restrict__ cat_ids,
                                    int64_t *__restrict__ merged_id,
                                    int32_t *__restrict__ cat)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t c = category_id[i];
    const int32_t m = fl_index_of(msrc, n_merge, c);
    if (m >= 0) c = mdst[m];
    merged_id[i] = c;
    cat[i] = fl_index_of(cat_ids, n_cat, c);
}

// split a 62-bit non-negative key into two 31-bit halves for two radix sorts
__global__ void fl_split_kernel(int64_t n, const int64_t *__restrict__ key,
                                const int32_t *__restrict__ order,
                                int32_t *__restrict__ lo, int32_t *__restrict__ hi)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t k = key[order ? order[i] : i];
    if (lo) lo[i] = (int32_t)(k & 0x7fffffff);
    if (hi) hi[i] = (int32_t)((uint64_t)k >> 31);
}

__global__ void fl_compose_kernel(int64_t n, const int32_t *__restrict__ outer,
                                  const int32_t *__restrict__ inner,
                                  int32_t *__restrict__ out)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = outer[inner[i]];
}

// trk[box] = run index of its track id; a track's boxes must share one video
__global__ void fl_track_of_kernel(int64_t n, const int32_t *__restrict__ order,
                                   const int32_t *__restrict__ run_id,
                                   const int32_t *__restrict__ run_start,
                                   const int64_t *__restrict__ video_id,
                                   int32_t *__restrict__ trk,
                                   int32_t *__restrict__ status)
{
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t d = order[p], r = run_id[p];
    trk[d] = r;
    if (video_id[d] != video_id[order[run_start[r]]]) atomicMin(&status[2], d);
}

// sort keys as exact doubles (negated: the sort is descending in "score"):
//   keep[d] = -(image's first-seen rank * M + ordinal)   list order
//   visit[d] = -(image's visiting rank * M + ordinal)     visiting order
// and the selection: kept box of a known category with 0 < area < inf whose
// image is visited.  Unselected boxes get track key INT32_MAX in trk_sel,
// dropped ones in trk_keep.
__global__ void fl_keys_kernel(int64_t n, const int32_t *__restrict__ img,
                               const int32_t *__restrict__ ordinal,
                               const uint8_t *__restrict__ dropped,
                               const int32_t *__restrict__ cat,
                               const double *__restrict__ area,
                               const int32_t *__restrict__ trk,
                               const int32_t *__restrict__ img_rank,
                               const int32_t *__restrict__ visit_rank,
                               const double *__restrict__ img_frame,
                               const int32_t *__restrict__ tl_pos, double M,
                               double *__restrict__ keep_key,
                               double *__restrict__ visit_key,
                               double *__restrict__ frame_key,
                               double *__restrict__ pos_key,
                               int32_t *__restrict__ trk_keep,
                               int32_t *__restrict__ trk_sel)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t im = img[i];
    const bool kept = !dropped[i] && im >= 0;
    const double o = (double)ordinal[i];
    keep_key[i] = kept ? -((double)img_rank[im] * M + o) : 0.0;
    const int32_t vr = kept ? visit_rank[im] : -1;
    const double a = area[i];
    const bool sel = kept && vr >= 0 && cat[i] >= 0 && a > 0 && a < INFINITY;
    visit_key[i] = sel ? -((double)vr * M + o) : 0.0;
    frame_key[i] = sel ? -img_frame[im] : 0.0;
    pos_key[i] = sel ? -(double)tl_pos[im] : 0.0;
    trk_keep[i] = kept ? trk[i] : INT32_MAX;
    trk_sel[i] = sel ? trk[i] : INT32_MAX;
}

// numpy's pairwise summation (np.add.reduce of a contiguous float64 array):
// blocks of <= 128 with eight running sums, halves above that
struct FlGather {
    const double *v;
    const int32_t *idx;
    __device__ __forceinline__ double operator()(int64_t i) const { return v[idx[i]]; }
};
__device__ double fl_pairwise(const FlGather &a, int64_t off, int64_t n)
{
    if (n < 8) {
        double r = 0.0;
        for (int64_t i = 0; i < n; i++) r += a(off + i);
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a(off + j);
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a(off + i + j);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a(off + i);
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return fl_pairwise(a, off, n2) + fl_pairwise(a, off + n2, n - n2);
}

// lane per track over its kept boxes (order_k lists them track by track in
// list order; kstart / kcount from the runs of the sorted track keys)
__global__ void fl_track_kept_kernel(int64_t n_runs, const int32_t *__restrict__ run_trk,
                                     const int32_t *__restrict__ run_start,
                                     int64_t n_sorted,
                                     const int32_t *__restrict__ order_k,
                                     const double *__restrict__ score,
                                     const int64_t *__restrict__ merged_id,
                                     double *__restrict__ trk_score,
                                     int32_t *__restrict__ trk_first_kept,
                                     int32_t *__restrict__ status)
{
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= n_runs) return;
    const int32_t t = run_trk[r];
    if (t == INT32_MAX) return;                   // the dropped boxes
    const int64_t b = run_start[r];
    const int64_t e = r + 1 < n_runs ? run_start[r + 1] : n_sorted;
    const int32_t first = order_k[b];
    const double s0 = score[first];
    const int64_t c0 = merged_id[first];
    bool same = true;
    for (int64_t i = b + 1; i < e; i++) {
        const int32_t d = order_k[i];
        if (score[d] != s0) same = false;
        if (merged_id[d] != c0) atomicMin(&status[3], d);
    }
    double s = s0;
    if (!same) {
        FlGather g{score, order_k};
        s = fl_pairwise(g, b, e - b) / (double)(e - b);      // np.mean
        status[1] = 1;                                       // required_average
    }
    trk_score[t] = s;
    trk_first_kept[t] = first;
}

// lane per track over its selected boxes in frame order (order_s)
__global__ void fl_track_sel_kernel(int64_t n_runs, const int32_t *__restrict__ run_trk,
                                    const int32_t *__restrict__ run_start,
                                    int64_t n_sorted,
                                    const int32_t *__restrict__ order_s,
                                    const double *__restrict__ area,
                                    const double *__restrict__ visit_key,
                                    const int32_t *__restrict__ img,
                                    double *__restrict__ sel_area,
                                    int32_t *__restrict__ sel_len,
                                    int32_t *__restrict__ sel_frames,
                                    double *__restrict__ sel_first)
{
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= n_runs) return;
    if (run_trk[r] == INT32_MAX) {
        sel_len[r] = 0;
        sel_frames[r] = 0;
        return;
    }
    const int64_t b = run_start[r];
    const int64_t e = r + 1 < n_runs ? run_start[r + 1] : n_sorted;
    double acc = 0.0, first = -INFINITY;
    int32_t frames = 0;
    for (int64_t i = b; i < e; i++) {
        const int32_t d = order_s[i];
        acc = acc + area[d];                        // sum(...) left to right
        first = fmax(first, visit_key[d]);          // keys are negated: max = earliest
        if (i + 1 == e || img[order_s[i + 1]] != img[d]) frames++;
    }
    sel_area[r] = acc / (double)(e - b);
    sel_len[r] = (int32_t)(e - b);
    sel_frames[r] = frames;
    sel_first[r] = first;
}

// per selected track: tables + federated filter; key = cat * n_vid + vid
__global__ void fl_track_filter_kernel(
    int64_t n_runs, const int32_t *__restrict__ run_trk,
    const int32_t *__restrict__ sel_len, const int32_t *__restrict__ trk_first_kept,
    const int32_t *__restrict__ cat, const int64_t *__restrict__ merged_id,
    const int64_t *__restrict__ video_id, const int64_t *__restrict__ track_id,
    int64_t n_vid, const int64_t *__restrict__ vid_ids, int64_t n_gkeys,
    const int32_t *__restrict__ gkeys, const int32_t *__restrict__ vid_row,
    const int64_t *__restrict__ neg_off, const int64_t *__restrict__ neg_val,
    const int64_t *__restrict__ nel_off, const int64_t *__restrict__ nel_val,
    int32_t *__restrict__ key, uint8_t *__restrict__ flags,
    int32_t *__restrict__ n_keep, int32_t *__restrict__ status)
{
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= n_runs) return;
    int32_t k = INT32_MAX;
    uint8_t f = 0;
    if (run_trk[r] != INT32_MAX && sel_len[r] > 0) {
        const int32_t first = trk_first_kept[run_trk[r]];
        const int32_t v = fl_index_of(vid_ids, n_vid, video_id[first]);
        const int32_t c = cat[first];
        if (v < 0) {
            atomicMin(&status[0], first);          // track of an unknown video
        } else {
            const int32_t kk = c * (int32_t)n_vid + v;
            const int32_t row = vid_row[v];
            const int64_t cid = merged_id[first];
            if (fl_in_sorted(gkeys, n_gkeys, kk) || fl_in_list(neg_off, neg_val, row, cid)) {
                k = kk;
                f = (fl_in_list(nel_off, nel_val, row, cid) ? 1 : 0) |
                    (track_id[first] <= 0 ? 2 : 0);
                atomicAdd(n_keep, 1);
            }
        }
    }
    key[r] = k;
    flags[r] = f;
}

// frame lists of the final tracks: one box per image, the last one in frame
// order; positions from the per-video timeline
__global__ void fl_frames_kernel(int64_t n_trk, const int32_t *__restrict__ trk_run,
                                 const int32_t *__restrict__ run_start,
                                 int64_t n_runs, int64_t n_sorted,
                                 const int32_t *__restrict__ order_s,
                                 const int32_t *__restrict__ img,
                                 const int32_t *__restrict__ tl_pos,
                                 const double4 *__restrict__ bbox,
                                 const int32_t *__restrict__ frame_off,
                                 int32_t *__restrict__ frame_pos,
                                 double4 *__restrict__ frame_box)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n_trk) return;
    const int32_t r = trk_run[t];
    const int64_t b = run_start[r];
    const int64_t e = (int64_t)r + 1 < n_runs ? run_start[r + 1] : n_sorted;
    int64_t w = frame_off[t];
    for (int64_t i = b; i < e; i++) {
        const int32_t d = order_s[i];
        if (i + 1 == e || img[order_s[i + 1]] != img[d]) {
            frame_pos[w] = tl_pos[img[d]];
            frame_box[w] = bbox[d];
            w++;
        }
    }
}

__global__ void fl_gather_i32_kernel(int64_t n, const int32_t *__restrict__ idx,
                                     const int32_t *__restrict__ src,
                                     int32_t *__restrict__ out)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}

__global__ void fl_gather_f64_kernel(int64_t n, const int32_t *__restrict__ idx,
                                     const double *__restrict__ src,
                                     double *__restrict__ out)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}

__global__ void fl_gather_u8_kernel(int64_t n, const int32_t *__restrict__ idx,
                                    const uint8_t *__restrict__ src,
                                    uint8_t *__restrict__ out)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}

#define FL_LAUNCH1(name, n, ...)                                              \
    TAO_TIMED(#name, s, name<<<fl_blocks(n, 256), 256, 0, s>>>(__VA_ARGS__))

extern "C" int taoamd_flat_ordinal(int64_t n, const int32_t *order, const int32_t *img,
                                   const int32_t *img_start, int32_t max_dets,
                                   int32_t *ordinal, uint8_t *dropped, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0) return TAOAMD_ERR_ARG;
    if (n == 0) return TAOAMD_OK;
    if (!order || !img || !img_start || !ordinal || !dropped) return TAOAMD_ERR_ARG;
    FL_LAUNCH1(fl_ordinal_kernel, n, n, order, img, img_start, max_dets, ordinal, dropped);
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_count_bad(int64_t n, const double *bbox,
                                     const uint8_t *dropped, int32_t *count,
                                     void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0 || !count) return TAOAMD_ERR_ARG;
    TAO_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), s));
    if (n == 0) return TAOAMD_OK;
    if (!bbox || !dropped) return TAOAMD_ERR_ARG;
    FL_LAUNCH1(fl_count_bad_kernel, n, n, (const double4 *)bbox, dropped, count);
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_ordscore(int64_t n, const int32_t *img,
                                    const int32_t *img_count, const double *score,
                                    int32_t max_dets, double *out, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0) return TAOAMD_ERR_ARG;
    if (n == 0) return TAOAMD_OK;
    if (!img || !img_count || !score || !out) return TAOAMD_ERR_ARG;
    FL_LAUNCH1(fl_ordscore_kernel, n, n, img, img_count, score, max_dets, out);
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_merge_cat(int64_t n, const int64_t *category_id,
                                     int64_t n_merge, const int64_t *merge_src,
                                     const int64_t *merge_dst, int64_t n_cat,
                                     const int64_t *cat_ids, int64_t *merged_id,
                                     int32_t *cat, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0 || n_merge < 0 || n_cat < 0) return TAOAMD_ERR_ARG;
    if (n == 0) return TAOAMD_OK;
    if (!category_id || !cat_ids || !merged_id || !cat || (n_merge && (!merge_src || !merge_dst)))
        return TAOAMD_ERR_ARG;
    FL_LAUNCH1(fl_merge_cat_kernel, n, n, category_id, n_merge, merge_src, merge_dst,
               n_cat, cat_ids, merged_id, cat);
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_split64(int64_t n, const int64_t *key, const int32_t *order,
                                   int32_t *lo, int32_t *hi, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0) return TAOAMD_ERR_ARG;
    if (n == 0) return TAOAMD_OK;
    if (!key) return TAOAMD_ERR_ARG;
    FL_LAUNCH1(fl_split_kernel, n, n, key, order, lo, hi);
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_compose(int64_t n, const int32_t *outer, const int32_t *inner,
                                   int32_t *out, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0) return TAOAMD_ERR_ARG;
    if (n == 0) return TAOAMD_OK;
    if (!outer || !inner || !out) return TAOAMD_ERR_ARG;
    FL_LAUNCH1(fl_compose_kernel, n, n, outer, inner, out);
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_gather_cols(int64_t n, const int32_t *idx,
                                       const int32_t *src_i32, int32_t *out_i32,
                                       const double *src_f64, double *out_f64,
                                       const uint8_t *src_u8, uint8_t *out_u8,
                                       void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0) return TAOAMD_ERR_ARG;
    if (n == 0) return TAOAMD_OK;
    if (!idx) return TAOAMD_ERR_ARG;
    if (src_i32 && out_i32) FL_LAUNCH1(fl_gather_i32_kernel, n, n, idx, src_i32, out_i32);
    if (src_f64 && out_f64) FL_LAUNCH1(fl_gather_f64_kernel, n, n, idx, src_f64, out_f64);
    if (src_u8 && out_u8) FL_LAUNCH1(fl_gather_u8_kernel, n, n, idx, src_u8, out_u8);
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_track_of(int64_t n, const int32_t *order, const int32_t *run_id,
                                    const int32_t *run_start, const int64_t *video_id,
                                    int32_t *trk, int32_t *status, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0) return TAOAMD_ERR_ARG;
    if (n == 0) return TAOAMD_OK;
    if (!order || !run_id || !run_start || !video_id || !trk || !status) return TAOAMD_ERR_ARG;
    FL_LAUNCH1(fl_track_of_kernel, n, n, order, run_id, run_start, video_id, trk, status);
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_keys(int64_t n, const int32_t *img, const int32_t *ordinal,
                                const uint8_t *dropped, const int32_t *cat,
                                const double *area, const int32_t *trk,
                                const int32_t *img_rank, const int32_t *visit_rank,
                                const double *img_frame, const int32_t *tl_pos,
                                double M, double *keep_key, double *visit_key,
                                double *frame_key, double *pos_key,
                                int32_t *trk_keep, int32_t *trk_sel, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0) return TAOAMD_ERR_ARG;
    if (n == 0) return TAOAMD_OK;
    if (!img || !ordinal || !dropped || !cat || !area || !trk || !img_rank ||
        !visit_rank || !img_frame || !tl_pos || !keep_key || !visit_key ||
        !frame_key || !pos_key || !trk_keep || !trk_sel)
        return TAOAMD_ERR_ARG;
    FL_LAUNCH1(fl_keys_kernel, n, n, img, ordinal, dropped, cat, area, trk, img_rank,
               visit_rank, img_frame, tl_pos, M, keep_key, visit_key, frame_key,
               pos_key, trk_keep, trk_sel);
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_track_kept(int64_t n_runs, const int32_t *run_trk,
                                      const int32_t *run_start, int64_t n_sorted,
                                      const int32_t *order_k, const double *score,
                                      const int64_t *merged_id, double *trk_score,
                                      int32_t *trk_first_kept, int32_t *status,
                                      void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n_runs < 0) return TAOAMD_ERR_ARG;
    if (n_runs == 0) return TAOAMD_OK;
    if (!run_trk || !run_start || !order_k || !score || !merged_id || !trk_score ||
        !trk_first_kept || !status)
        return TAOAMD_ERR_ARG;
    FL_LAUNCH1(fl_track_kept_kernel, n_runs, n_runs, run_trk, run_start, n_sorted,
               order_k, score, merged_id, trk_score, trk_first_kept, status);
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_track_sel(int64_t n_runs, const int32_t *run_trk,
                                     const int32_t *run_start, int64_t n_sorted,
                                     const int32_t *order_s, const double *area,
                                     const double *visit_key, const int32_t *img,
                                     double *sel_area, int32_t *sel_len,
                                     int32_t *sel_frames, double *sel_first,
                                     void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n_runs < 0) return TAOAMD_ERR_ARG;
    if (n_runs == 0) return TAOAMD_OK;
    if (!run_trk || !run_start || !order_s || !area || !visit_key || !img ||
        !sel_area || !sel_len || !sel_frames || !sel_first)
        return TAOAMD_ERR_ARG;
    FL_LAUNCH1(fl_track_sel_kernel, n_runs, n_runs, run_trk, run_start, n_sorted,
               order_s, area, visit_key, img, sel_area, sel_len, sel_frames, sel_first);
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_track_filter(
    int64_t n_runs, const int32_t *run_trk, const int32_t *sel_len,
    const int32_t *trk_first_kept, const int32_t *cat, const int64_t *merged_id,
    const int64_t *video_id, const int64_t *track_id, int64_t n_vid,
    const int64_t *vid_ids, int64_t n_gkeys, const int32_t *gkeys,
    const int32_t *vid_row, const int64_t *neg_off, const int64_t *neg_val,
    const int64_t *nel_off, const int64_t *nel_val, int32_t *key, uint8_t *flags,
    int32_t *n_keep, int32_t *status, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n_runs < 0 || !n_keep) return TAOAMD_ERR_ARG;
    TAO_HIP(hipMemsetAsync(n_keep, 0, sizeof(int32_t), s));
    if (n_runs == 0) return TAOAMD_OK;
    if (!run_trk || !sel_len || !trk_first_kept || !cat || !merged_id || !video_id ||
        !track_id || !vid_ids || !vid_row || !key || !flags || !status ||
        (n_gkeys > 0 && !gkeys))
        return TAOAMD_ERR_ARG;
    FL_LAUNCH1(fl_track_filter_kernel, n_runs, n_runs, run_trk, sel_len, trk_first_kept,
               cat, merged_id, video_id, track_id, n_vid, vid_ids, n_gkeys, gkeys,
               vid_row, neg_off, neg_val, nel_off, nel_val, key, flags, n_keep, status);
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_flat_frames(int64_t n_trk, const int32_t *trk_run,
                                  const int32_t *run_start, int64_t n_runs,
                                  int64_t n_sorted, const int32_t *order_s,
                                  const int32_t *img, const int32_t *tl_pos,
                                  const double *bbox, const int32_t *frame_off,
                                  int32_t *frame_pos, double *frame_box, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n_trk < 0) return TAOAMD_ERR_ARG;
    if (n_trk == 0) return TAOAMD_OK;
    if (!trk_run || !run_start || !order_s || !img || !tl_pos || !bbox || !frame_off ||
        !frame_pos || !frame_box)
        return TAOAMD_ERR_ARG;
    FL_LAUNCH1(fl_frames_kernel, n_trk, n_trk, trk_run, run_start, n_runs, n_sorted,
               order_s, img, tl_pos, (const double4 *)bbox, frame_off, frame_pos,
               (double4 *)frame_box);
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

// exclusive scan of int32 counts (single workgroup) -- frame offsets
extern "C" int taoamd_flat_scan(int64_t n, const int32_t *count, int32_t *start,
                                int32_t *status2, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n < 0 || !start || !status2 || (n && !count)) return TAOAMD_ERR_ARG;
    TAO_TIMED("fl_starts_kernel", s, fl_starts_kernel<<<1, 1024, 0, s>>>(
        n, count, start, status2));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}
