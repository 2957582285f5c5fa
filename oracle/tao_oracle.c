/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into or called by the
 * product path (only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load this library).
 *
 * Plain-C restatement, on the flattened cell tables, of the arithmetic the
 * reference evaluator runs per cell and per (category, range):
 *
 *   orc_bb_iou          bbIou, iscrowd == 0   (vendored pycocotools
 *                       common/maskApi.c:109-120; called from
 *                       tao_amodal/evaluation/lvis_amodal/eval.py:191)
 *   orc_lvis_ranges     GT ignore flag per visibility range
 *                       (lvis_amodal/eval.py:202-217) and the unmatched-
 *                       detection ignore mask (:281-286)
 *   orc_tao_ranges      the same for the 5 area x 4 duration ranges
 *                       (tao_amodal/evaluation/tao_amodal/eval.py:348-368,
 *                       432-439)
 *   orc_track_iou       3D IoU of every (dt track, gt track) pair of a cell
 *                       (tao_amodal/eval.py:15-48,73-96), frames summed in
 *                       ascending timeline order (see oracle/pyoracle.py on
 *                       frame_order)
 *   orc_match           greedy assignment at 10 IoU thresholds for every
 *                       range (lvis_amodal/eval.py:219-290 ==
 *                       tao_amodal/eval.py:370-443)
 *   orc_accumulate      stable score sort per category, TP/FP sweep,
 *                       precision envelope, 101-point recall sampling
 *                       (lvis_amodal/eval.py:339-417 ==
 *                       tao_amodal/eval.py:496-573)
 *
 * Parity status: PINNED -- tests/test_flat_oracle_golden.py checks every
 * function against the golden vectors generated from the reference
 * (tests/golden/, fixtures F1-F5).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no fast-math).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define N_THR 10
#define N_REC 101
#define MAX_RNG 32

#define GT_IGNORE 1
#define GT_OOF 2
#define GT_ID_HIDDEN 4
#define DT_IGNORE_UNMATCHED 1
#define DT_NO_CONSUME 2

/* np.linspace(0.5, 0.95, 10) and np.linspace(0, 1, 101), bit for bit
 * (lvis_amodal/eval.py:560-565).  linspace computes start + i*step with
 * step = (stop-start)/div; the last point is forced to stop. */
void orc_thresholds(double *iou_thrs, double *rec_thrs)
{
    double step = (0.95 - 0.5) / 9.0;
    for (int i = 0; i < N_THR; i++)
        iou_thrs[i] = 0.5 + (double)i * step;
    iou_thrs[N_THR - 1] = 0.95;
    step = (1.0 - 0.0) / 100.0;
    for (int i = 0; i < N_REC; i++)
        rec_thrs[i] = 0.0 + (double)i * step;
    rec_thrs[N_REC - 1] = 1.0;
}

static double box_iou(const double *D, const double *G)
{
    double da = D[2] * D[3], ga = G[2] * G[3];
    double w = fmin(D[2] + D[0], G[2] + G[0]) - fmax(D[0], G[0]);
    if (w <= 0)
        return 0;
    double h = fmin(D[3] + D[1], G[3] + G[1]) - fmax(D[1], G[1]);
    if (h <= 0)
        return 0;
    double i = w * h;
    double u = da + ga - i;
    return i / u;
}

/* o[g*m+d], the layout of bbIou */
void orc_bb_iou(const double *dt, const double *gt, size_t m, size_t n,
                double *o)
{
    for (size_t g = 0; g < n; g++)
        for (size_t d = 0; d < m; d++)
            o[g * m + d] = box_iou(dt + 4 * d, gt + 4 * g);
}

/* bit r of gt_rng[g]: GT g is ignored in range r; bit r of dt_rng[d]: an
 * unmatched detection d is ignored in range r */
void orc_lvis_ranges(int64_t n_gt, const double *gt_vis,
                     const uint8_t *gt_flags, int64_t n_dt,
                     const uint8_t *dt_flags, uint32_t *gt_rng,
                     uint32_t *dt_rng)
{
    static const double lo[5] = {0, 0, 0.1, 0.8, 0};
    static const double hi[5] = {1.0, 0.1, 0.8, 1.0, 0.8};
    for (int64_t g = 0; g < n_gt; g++) {
        uint32_t m = 0;
        int ign = gt_flags[g] & GT_IGNORE;
        for (int r = 0; r < 5; r++)
            if (ign || gt_vis[g] < lo[r] || gt_vis[g] > hi[r])
                m |= 1u << r;
        if (ign || !(gt_flags[g] & GT_OOF))
            m |= 1u << 5;
        gt_rng[g] = m;
    }
    for (int64_t d = 0; d < n_dt; d++)
        dt_rng[d] = (dt_flags[d] & DT_IGNORE_UNMATCHED) ? 0x3fu : 0u;
}

void orc_tao_ranges(int64_t n_gt, const double *gt_area,
                    const int32_t *gt_len, const int32_t *gt_nhp,
                    const uint8_t *gt_flags, int64_t n_dt,
                    const double *dt_area, const int32_t *dt_len,
                    const uint8_t *dt_flags, uint32_t *gt_rng,
                    uint32_t *dt_rng)
{
    static const double alo[5] = {0, 0, 1024, 9216, 0};
    static const double ahi[5] = {1e10, 1024, 9216, 1e10, 1e10};
    static const double tlo[4] = {0, 0, 3, 10};
    static const double thi[4] = {1e5, 3, 10, 1e5};
    for (int64_t g = 0; g < n_gt; g++) {
        uint32_t m = 0;
        int ign = gt_flags[g] & GT_IGNORE;
        double len = (double)gt_len[g];
        for (int a = 0; a < 5; a++)
            for (int t = 0; t < 4; t++) {
                int bad = ign || gt_area[g] < alo[a] || gt_area[g] > ahi[a] ||
                          len < tlo[t] || len > thi[t];
                if (a == 4 && gt_nhp[g] <= 5)
                    bad = 1;
                if (bad)
                    m |= 1u << (a * 4 + t);
            }
        gt_rng[g] = m;
    }
    for (int64_t d = 0; d < n_dt; d++) {
        uint32_t m = 0;
        int nel = dt_flags[d] & DT_IGNORE_UNMATCHED;
        double len = (double)dt_len[d];
        for (int a = 0; a < 5; a++)
            for (int t = 0; t < 4; t++)
                if (nel || dt_area[d] < alo[a] || dt_area[d] > ahi[a] ||
                    len < tlo[t] || len > thi[t])
                    m |= 1u << (a * 4 + t);
        dt_rng[d] = m;
    }
}

/* iou[cell_iou_off[c] + d*G + g]; returns the number of per-frame box pairs
 * evaluated (the P_T of SURVEY.md 8(d)).
 * mode 0: 3d_iou       sum_f inter / sum_f union      (eval.py:73-96)
 * mode 1: avg_iou      mean_f (inter_f / union_f)      (eval.py:99-117), sum
 *                      taken left to right in timeline order
 * mode 2: imagenetvid  #{f: inter_f > 0.5 union_f} / #frames (eval.py:51-70) */
/* Threads used by the cell / category loops below (OpenMP).  1 = the plain
 * scalar port (default); bench.py's all-cores baseline raises it.  Cells and
 * categories are independent, so the results do not depend on it. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int orc_max_threads(void) { return omp_get_max_threads(); }

int64_t orc_track_iou(int64_t n_cells, const int32_t *cell_dt_off,
                      const int32_t *cell_gt_off, const int64_t *cell_iou_off,
                      const int32_t *dt_foff, const int32_t *dt_fpos,
                      const double *dt_fbox, const int32_t *gt_foff,
                      const int32_t *gt_fpos, const double *gt_fbox,
                      int mode, double *iou)
{
    int64_t pairs = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : pairs) num_threads(g_threads)
    for (int64_t c = 0; c < n_cells; c++) {
        int32_t d0 = cell_dt_off[c], D = cell_dt_off[c + 1] - d0;
        int32_t g0 = cell_gt_off[c], G = cell_gt_off[c + 1] - g0;
        for (int32_t d = 0; d < D; d++)
            for (int32_t g = 0; g < G; g++) {
                int32_t pd = dt_foff[d0 + d], ed = dt_foff[d0 + d + 1];
                int32_t pg = gt_foff[g0 + g], eg = gt_foff[g0 + g + 1];
                double i = 0, u = 0, acc = 0, cnt = 0;
                while (pd < ed || pg < eg) {
                    cnt += 1;
                    int32_t fd = pd < ed ? dt_fpos[pd] : INT32_MAX;
                    int32_t fg = pg < eg ? gt_fpos[pg] : INT32_MAX;
                    if (fd == fg) {
                        const double *B = dt_fbox + 4 * (int64_t)pd;
                        const double *A = gt_fbox + 4 * (int64_t)pg;
                        double w = fmin(B[0] + B[2], A[0] + A[2]) -
                                   fmax(B[0], A[0]);
                        double h = fmin(B[1] + B[3], A[1] + A[3]) -
                                   fmax(B[1], A[1]);
                        w = w > 0 ? w : 0;
                        h = h > 0 ? h : 0;
                        double i_ = w * h;
                        double u_ = B[2] * B[3] + A[2] * A[3] - i_;
                        i += i_;
                        u += u_;
                        if (mode == 1) acc += u_ > 0 ? i_ / u_ : 0;
                        if (mode == 2 && i_ > 0.5 * u_) acc += 1;
                        pd++, pg++, pairs++;
                    } else if (fg < fd) {
                        u += gt_fbox[4 * (int64_t)pg + 2] *
                             gt_fbox[4 * (int64_t)pg + 3];
                        pg++;
                    } else {
                        u += dt_fbox[4 * (int64_t)pd + 2] *
                             dt_fbox[4 * (int64_t)pd + 3];
                        pd++;
                    }
                }
                iou[cell_iou_off[c] + (int64_t)d * G + g] =
                    mode == 0 ? (u > 0 ? i / u : 0) : acc / cnt;
            }
    }
    return pairs;
}

/*
 * Greedy assignment for every cell, range and threshold.
 *
 *   boxes != NULL : LVIS, IoU computed from dt_box/gt_box per cell
 *   boxes == NULL : TAO, IoU read from iou[cell_iou_off[c] + d*G + g]
 *
 * Outputs, per detection d and word w (n_words = ceil(n_rng*10/64)), with
 * combo index c = r*10 + t:
 *   matched[d*n_words + w]  bit c%64: d got a GT whose id is visible
 *   ignored[d*n_words + w]  bit c%64: d is ignored
 *   match_gt[d*n_rng*10 + c] (optional) in-cell index of the GT taken, or -1
 *   ious_out (optional, LVIS): ious[cell_iou_off[c] + d*G + g]
 */
void orc_match(int64_t n_cells, const int32_t *cell_dt_off,
               const int32_t *cell_gt_off, const int64_t *cell_iou_off,
               const double *dt_box, const double *gt_box, const double *iou,
               int n_rng, const uint32_t *gt_rng, const uint32_t *dt_rng,
               const uint8_t *gt_flags, const uint8_t *dt_flags,
               uint64_t *matched, uint64_t *ignored, int32_t *match_gt,
               double *ious_out)
{
    double thr[N_THR], rec[N_REC];
    orc_thresholds(thr, rec);
    int n_combo = n_rng * N_THR;
    int n_words = (n_combo + 63) / 64;
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int64_t c = 0; c < n_cells; c++) {
        int32_t d0 = cell_dt_off[c], D = cell_dt_off[c + 1] - d0;
        int32_t g0 = cell_gt_off[c], G = cell_gt_off[c + 1] - g0;
        for (int32_t d = 0; d < D; d++)
            for (int w = 0; w < n_words; w++) {
                matched[(int64_t)(d0 + d) * n_words + w] = 0;
                ignored[(int64_t)(d0 + d) * n_words + w] = 0;
            }
        double *tile = NULL;
        const double *M;
        if (iou) {
            M = iou + cell_iou_off[c];
        } else {
            tile = (double *)malloc(sizeof(double) * (size_t)(D > 0 ? D : 1) *
                                    (size_t)(G > 0 ? G : 1));
            for (int32_t d = 0; d < D; d++)
                for (int32_t g = 0; g < G; g++)
                    tile[(int64_t)d * G + g] =
                        box_iou(dt_box + 4 * (int64_t)(d0 + d),
                                gt_box + 4 * (int64_t)(g0 + g));
            M = tile;
            if (ious_out)
                memcpy(ious_out + cell_iou_off[c], tile,
                       sizeof(double) * (size_t)D * (size_t)G);
        }
        uint8_t *taken = (uint8_t *)malloc((size_t)(G > 0 ? G : 1));
        for (int r = 0; r < n_rng; r++)
            for (int t = 0; t < N_THR; t++) {
                int combo = r * N_THR + t;
                memset(taken, 0, (size_t)(G > 0 ? G : 1));
                double start = thr[t] < 1 - 1e-10 ? thr[t] : 1 - 1e-10;
                for (int32_t d = 0; d < D; d++) {
                    double best = start;
                    int32_t m = -1;
                    /* evaluated GTs first, in visiting order (the stable
                     * ignore-last sort of lvis_amodal/eval.py:220) ... */
                    for (int32_t g = 0; g < G; g++) {
                        if ((gt_rng[g0 + g] >> r) & 1u) continue;
                        if (taken[g]) continue;
                        if (M[(int64_t)d * G + g] < best) continue;
                        best = M[(int64_t)d * G + g];
                        m = g;
                    }
                    /* ... ignored ones only when none of those matched */
                    if (m == -1)
                        for (int32_t g = 0; g < G; g++) {
                            if (!((gt_rng[g0 + g] >> r) & 1u)) continue;
                            if (taken[g]) continue;
                            if (M[(int64_t)d * G + g] < best) continue;
                            best = M[(int64_t)d * G + g];
                            m = g;
                        }
                    int vis = 0, ig = 0;
                    if (m >= 0) {
                        if (!(dt_flags[d0 + d] & DT_NO_CONSUME))
                            taken[m] = 1;
                        vis = !(gt_flags[g0 + m] & GT_ID_HIDDEN);
                        ig = (gt_rng[g0 + m] >> r) & 1u;
                    }
                    if (!vis && ((dt_rng[d0 + d] >> r) & 1u))
                        ig = 1;
                    int64_t wi = (int64_t)(d0 + d) * n_words + combo / 64;
                    if (vis) matched[wi] |= 1ull << (combo % 64);
                    if (ig) ignored[wi] |= 1ull << (combo % 64);
                    if (match_gt)
                        match_gt[(int64_t)(d0 + d) * n_combo + combo] = m;
                }
            }
        free(taken);
        free(tile);
    }
}

/* merge sort of indices by (cat asc, score desc), stable */
static const int32_t *s_cat;
static const double *s_score;
static int before(int64_t a, int64_t b)
{
    if (s_cat[a] != s_cat[b])
        return s_cat[a] < s_cat[b];
    return s_score[a] > s_score[b]; /* -score ascending */
}
static void msort(int64_t *x, int64_t *tmp, int64_t n)
{
    if (n < 2) return;
    int64_t h = n / 2;
#pragma omp task if (n > 65536)
    msort(x, tmp, h);
#pragma omp task if (n > 65536)
    msort(x + h, tmp + h, n - h);
#pragma omp taskwait
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < n)
        tmp[k++] = before(x[j], x[i]) ? x[j++] : x[i++];
    while (i < h) tmp[k++] = x[i++];
    while (j < n) tmp[k++] = x[j++];
    memcpy(x, tmp, sizeof(int64_t) * (size_t)n);
}

/*
 * precision[T][R][K][n_rng], recall[T][K][n_rng] (C order, -1 = absent).
 * Detections must be listed in concatenation order (cell by cell, score
 * order inside a cell).  Optional outputs: order[n_dt] (sorted position ->
 * detection), num_gt[K*n_rng].
 */
void orc_accumulate(int64_t n_dt, int32_t n_cat, int n_rng,
                    const int32_t *dt_cat, const double *dt_score,
                    const uint64_t *matched, const uint64_t *ignored,
                    int64_t n_gt, const int32_t *gt_cat,
                    const uint32_t *gt_rng, double *precision, double *recall,
                    int64_t *order_out, int32_t *num_gt_out)
{
    double thr[N_THR], rec[N_REC];
    orc_thresholds(thr, rec);
    int n_combo = n_rng * N_THR;
    int n_words = (n_combo + 63) / 64;
    int64_t KA = (int64_t)n_cat * n_rng;
    for (int64_t i = 0; i < (int64_t)N_THR * N_REC * KA; i++) precision[i] = -1;
    for (int64_t i = 0; i < (int64_t)N_THR * KA; i++) recall[i] = -1;
    int32_t *num_gt = (int32_t *)calloc((size_t)(KA > 0 ? KA : 1), 4);
    for (int64_t g = 0; g < n_gt; g++)
        for (int r = 0; r < n_rng; r++)
            if (!((gt_rng[g] >> r) & 1u))
                num_gt[(int64_t)gt_cat[g] * n_rng + r]++;
    if (num_gt_out) memcpy(num_gt_out, num_gt, sizeof(int32_t) * (size_t)KA);
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_dt + 1));
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_dt + 1));
    for (int64_t i = 0; i < n_dt; i++) order[i] = i;
    s_cat = dt_cat;
    s_score = dt_score;
#pragma omp parallel num_threads(g_threads)
#pragma omp single
    msort(order, tmp, n_dt);
    if (order_out) memcpy(order_out, order, sizeof(int64_t) * (size_t)n_dt);
    const double eps = 2.220446049250313e-16; /* np.spacing(1) */
    /* categories without detections still get precision 0 / recall 0 when
     * they have evaluated GT, hence the walk over all categories */
    int64_t *cat_b = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_cat + 1));
    {
        int64_t pos = 0;
        for (int32_t k = 0; k < n_cat; k++) {
            cat_b[k] = pos;
            while (pos < n_dt && dt_cat[order[pos]] == k) pos++;
        }
        cat_b[n_cat] = pos;
    }
    int64_t longest = 0;
    for (int32_t k = 0; k < n_cat; k++)
        if (cat_b[k + 1] - cat_b[k] > longest) longest = cat_b[k + 1] - cat_b[k];
#pragma omp parallel num_threads(g_threads)
    {
    double *pr = (double *)malloc(sizeof(double) * (size_t)(longest + 1));
    double *rc = (double *)malloc(sizeof(double) * (size_t)(longest + 1));
#pragma omp for schedule(dynamic, 4)
    for (int32_t k = 0; k < n_cat; k++) {
        int64_t b = cat_b[k];
        int64_t n = cat_b[k + 1] - b;
        for (int r = 0; r < n_rng; r++) {
            int32_t ng = num_gt[(int64_t)k * n_rng + r];
            if (ng == 0) continue;
            for (int t = 0; t < N_THR; t++) {
                int combo = r * N_THR + t;
                double tp = 0, fp = 0;
                for (int64_t i = 0; i < n; i++) {
                    int64_t d = order[b + i];
                    int mt = (matched[d * n_words + combo / 64] >>
                              (combo % 64)) & 1;
                    int ig = (ignored[d * n_words + combo / 64] >>
                              (combo % 64)) & 1;
                    if (mt && !ig) tp += 1;
                    if (!mt && !ig) fp += 1;
                    rc[i] = tp / (double)ng;
                    pr[i] = tp / (fp + tp + eps);
                }
                recall[((int64_t)t * n_cat + k) * n_rng + r] =
                    n ? rc[n - 1] : 0;
                for (int64_t i = n - 1; i > 0; i--)
                    if (pr[i] > pr[i - 1]) pr[i - 1] = pr[i];
                int64_t idx = 0;
                for (int j = 0; j < N_REC; j++) {
                    while (idx < n && rc[idx] < rec[j]) idx++; /* 'left' */
                    double v = idx < n ? pr[idx] : 0.0;
                    precision[(((int64_t)t * N_REC + j) * n_cat + k) * n_rng +
                              r] = v;
                }
            }
        }
    }
    free(pr); free(rc);
    }
    free(cat_b); free(order); free(tmp); free(num_gt);
}
