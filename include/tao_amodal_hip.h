/*
 * tao_amodal_hip.h -- C ABI of the MI355X (gfx950) evaluation hot path.
 *
 * This is the drop-in boundary for the path the reference runs in
 * tools/eval_on_tao_amodal.py (LVISEval + TaoEval).  The reference has exactly
 * one native seam on that path, pycocotools' bbIou; everything else is Python
 * that this library replaces with batched device kernels over flattened cell
 * tables.  Every entry point is extern "C", takes plain pointers and sizes,
 * returns an int status (0 = ok, see taoamd_strerror) and never throws.  All
 * pointers are DEVICE pointers unless the name ends in _host; the caller owns
 * every buffer; `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  Calls are asynchronous on `stream` unless stated otherwise.  One
 * host thread per device.
 *
 * Reference interfaces replaced (paths relative to /root/reference;
 * L/ = tao_amodal/evaluation/lvis_amodal/, T/ = tao_amodal/evaluation/
 * tao_amodal/, C/ = visualization/tao/third_party/pysot/training_dataset/
 * coco/pycocotools/common/):
 *
 *   taoamd_bb_iou[_host]   void bbIou(BB dt, BB gt, siz m, siz n, byte
 *                          *iscrowd, double *o)            C/maskApi.h:41-42,
 *                          C/maskApi.c:109-120, called at L/eval.py:191
 *   taoamd_lvis_ranges     LVISEval.evaluate_img, GT _ignore per visibility
 *                          range + dt_ig_mask              L/eval.py:202-217,
 *                                                          281-288
 *   taoamd_tao_ranges      TaoEval.evaluate_vid, the same for 5 area x 4
 *                          duration ranges                 T/eval.py:348-368,
 *                                                          432-441
 *   taoamd_track_iou       TaoEval.compute_iou / compute_track_box_iou /
 *                          bb_intersect_union              T/eval.py:15-48,
 *                                                          73-96,306-335
 *   taoamd_match           LVISEval.compute_iou + evaluate_img greedy loop,
 *                          TaoEval.evaluate_vid greedy loop
 *                                                          L/eval.py:168-192,
 *                                                          219-290;
 *                                                          T/eval.py:370-443
 *   taoamd_sort_by_cat_score  np.argsort(-dt_scores, kind="mergesort") per
 *                          category                        L/eval.py:353-361,
 *                                                          T/eval.py:508-518
 *   taoamd_accumulate      LVISEval.accumulate / TaoEval.accumulate
 *                                                          L/eval.py:339-426,
 *                                                          T/eval.py:496-584
 *
 * Data model ("cell tables", built by tao_amodal_amd/flatten.py):
 *   a cell = one (image, category) [LVIS] or (video, category) [TAO] pair that
 *   has at least one detection or ground truth.  Detections (LVIS) / detection
 *   tracks (TAO) of a cell are contiguous and sorted by descending score
 *   (stable); ground truths are contiguous in the reference's visiting order.
 *   cell_dt_off / cell_gt_off are CSR offsets (int32, n_cells + 1 entries).
 *
 *   A "combo" is one (range r, IoU threshold t) pair, combo = r*10 + t;
 *   n_rng = 6 (LVIS visibility ranges) or 20 (TAO area x duration, r = a*4+t_).
 *   Per detection the kernels emit n_words = ceil(n_rng*10/64) 64-bit words
 *   of `matched` bits and of `ignored` bits (bit combo%64 of word combo/64):
 *     TP = matched & ~ignored, FP = ~matched & ~ignored.
 */
#ifndef TAO_AMODAL_HIP_H
#define TAO_AMODAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAOAMD_N_THR 10   /* IoU thresholds 0.50:0.05:0.95 */
#define TAOAMD_N_REC 101  /* recall thresholds 0:0.01:1 */
#define TAOAMD_LVIS_RNG 6
#define TAOAMD_TAO_RNG 20

/* per-ground-truth flag bits */
#define TAOAMD_GT_IGNORE 1     /* truthy "ignore" field */
#define TAOAMD_GT_OOF 2        /* annotation is out_of_frame (LVIS side) */
#define TAOAMD_GT_ID_HIDDEN 4  /* id equals the "unmatched" sentinel (0 / -1) */
/* per-detection flag bits */
#define TAOAMD_DT_IGNORE_UNMATCHED 1 /* not-exhaustive category / area window */
#define TAOAMD_DT_NO_CONSUME 2       /* id <= 0: a match leaves the GT free */

/* status codes */
#define TAOAMD_OK 0
#define TAOAMD_ERR_HIP 1         /* a HIP runtime call failed */
#define TAOAMD_ERR_ARG 2         /* bad argument */
#define TAOAMD_ERR_TOO_LARGE 3   /* a cell exceeds a kernel limit */
#define TAOAMD_ERR_WORKSPACE 4   /* workspace too small */
#define TAOAMD_JSON_FALLBACK 16  /* taoamd_json_pred_open: the host reader should take the file */

const char *taoamd_strerror(int status);
/* text of the last failing HIP call on this thread ("" if none) */
const char *taoamd_last_error(void);
int taoamd_version(void);

/* Per-kernel timing for bench.py's roofline: while enabled, every kernel the
 * library launches is bracketed by two HIP events recorded on the stream it
 * is launched on (costs two event records per launch; off by default).
 * taoamd_kernel_timing_collect waits for the recorded events, adds up the
 * launch durations by kernel name and forgets them: names = NUL-separated
 * list in launch order of first appearance, total_ms / calls per name;
 * *n_kernels = number of distinct names.  A call always consumes the records
 * (give room for 64 names).  Host pointers; synchronous. */
int taoamd_kernel_timing_enable(int on);
/* prefix ("label:") of the names recorded by this thread from now on, e.g. the
 * evaluator a pass belongs to; NULL or "" = none */
int taoamd_kernel_timing_label(const char *label);
int taoamd_kernel_timing_collect(char *names_host, size_t names_bytes,
                                 double *total_ms_host, int64_t *calls_host,
                                 int32_t max_kernels, int32_t *n_kernels_host);

/* Exact bit patterns of np.linspace(.5,.95,10) / np.linspace(0,1,101)
 * (L/eval.py:560-565).  Host pointers. */
int taoamd_thresholds_host(double *iou_thrs, double *rec_thrs);

/* The evaluation constants a caller of the class API may edit before run()
 * (`params.iou_thrs`, `rec_thrs`: L/eval.py:234,319-322,407, T/eval.py:385,
 * 473-477,562; `visibility_rng`: L/eval.py:143,205; `area_rng` / `time_rng`:
 * T/eval.py:272-275,348-368,432-441).  They reach the kernels as by-value
 * arguments; these two calls replace the CALLING THREAD's tables for its later
 * launches (NULL = the reference's default for that table).  The counts are
 * the kernels': TAOAMD_N_THR thresholds, TAOAMD_N_REC recall thresholds -- both
 * ASCENDING (equal neighbours allowed; a caller with fewer pads with copies of
 * its last value, one with more or in another order runs blocks and permutes:
 * evaluation/_core.py), else TAOAMD_ERR_ARG --, 5 visibility ranges (the sixth
 * range of the image level is the out-of-frame one and has no bounds), 5 area
 * ranges (the last one is the occlusion range: T/eval.py:272) and 4 duration
 * ranges, each as {lo, hi} pairs, bounds inclusive.  Host pointers. */
int taoamd_set_thresholds(const double *iou_thrs, const double *rec_thrs);
int taoamd_set_ranges(const double *visibility_rng, const double *area_rng,
                      const double *time_rng);

/* ---- bbIou ---------------------------------------------------------------
 * o[g*m + d] = IoU(dt[d], gt[g]); boxes are x,y,w,h doubles.  iscrowd may be
 * NULL (all zero); a non-zero entry selects u = area(dt) as in bbIou. */
int taoamd_bb_iou(const double *dt, const double *gt, size_t m, size_t n,
                  const unsigned char *iscrowd, double *o, void *stream);
/* same, host pointers in and out (allocates, copies, synchronises) */
int taoamd_bb_iou_host(const double *dt, const double *gt, size_t m, size_t n,
                       const unsigned char *iscrowd, double *o);

/* ---- range masks ------------------------------------------------------------
 * gt_rng[g] bit r: GT g is ignored in range r.  dt_rng[d] bit r: detection d
 * is ignored in range r when it ends up unmatched.  num_gt[k*n_rng + r]
 * (int32, written by the call) counts the evaluated GTs of category k.
 * gt_cat_off (optional, int32[n_cat+1], device): when the GTs are grouped by
 * category (category-major cell tables) the counts are formed by one
 * wavefront per category with ballots -- no atomics; NULL selects the
 * atomicAdd histogram over gt_cat.
 * Image level: a detection's mask is "every range" iff TAOAMD_DT_IGNORE_UNMATCHED
 * is set in dt_flags and 0 otherwise, so the table is optional -- dt_rng NULL in
 * taoamd_lvis_ranges skips it, dt_rng NULL in taoamd_match derives the mask
 * from dt_flags (240 MB less traffic per pass at 30 M detections). */
int taoamd_lvis_ranges(int64_t n_gt, const double *gt_vis,
                       const uint8_t *gt_flags, const int32_t *gt_cat,
                       const int32_t *gt_cat_off, int64_t n_dt,
                       const uint8_t *dt_flags, int32_t n_cat,
                       uint32_t *gt_rng, uint32_t *dt_rng, int32_t *num_gt,
                       void *stream);
int taoamd_tao_ranges(int64_t n_gt, const double *gt_area,
                      const int32_t *gt_len, const int32_t *gt_nhp,
                      const uint8_t *gt_flags, const int32_t *gt_cat,
                      const int32_t *gt_cat_off, int64_t n_dt,
                      const double *dt_area, const int32_t *dt_len,
                      const uint8_t *dt_flags, int32_t n_cat,
                      uint32_t *gt_rng, uint32_t *dt_rng, int32_t *num_gt,
                      void *stream);

/* ---- IoU of run-length masks (iou_type="segm") -------------------------------
 * iou[cell_iou_off[c] + d*G + g] for every cell c, iscrowd = 0: replaces
 * mask_utils.iou(dt, gt, iscrowd) on RLE objects (L/eval.py:179-191 ->
 * C/maskApi.c rleIou:77-96).  Masks are CSR run lists (column-major run
 * lengths, zeros first) with their frame size hw[n][2] = (height, width) and
 * tight box bb[n][4] = (x, y, w, h) as taoamd_rle_copy (tao_amodal_ingest.h)
 * produces them; row i belongs to detection / ground truth i of the cell
 * tables (n_dt / n_gt rows, dt_total / gt_total runs in all).  0 where the
 * tight boxes do not overlap, -1 where the frames differ, else |d & g| / |d | g|
 * (an empty intersection over a union of 1).  Workspace:
 * taoamd_rle_iou_workspace(n_dt, dt_total, n_gt, gt_total) bytes (per-run
 * prefix sums, rebuilt by every call). */
size_t taoamd_rle_iou_workspace(int64_t n_dt, int64_t dt_total, int64_t n_gt,
                                int64_t gt_total);
int taoamd_rle_iou(int64_t n_cells, const int32_t *cell_dt_off,
                   const int32_t *cell_gt_off, const int64_t *cell_iou_off,
                   int64_t n_dt, int64_t dt_total, const int64_t *dt_off,
                   const uint32_t *dt_runs, const int32_t *dt_hw,
                   const double *dt_bb, int64_t n_gt, int64_t gt_total,
                   const int64_t *gt_off, const uint32_t *gt_runs,
                   const int32_t *gt_hw, const double *gt_bb, double *iou,
                   void *workspace, size_t workspace_bytes, void *stream);

/* ---- 3D IoU of track pairs --------------------------------------------------
 * iou[cell_iou_off[c] + d*G + g] for every cell c (G = its GT track count).
 * Tracks are CSR lists of (timeline position, box) sorted by position.
 * pair_frames (optional, int64[1], zeroed by the call) receives the number of
 * same-frame box pairs evaluated (the unit of BASELINE.json's metric).
 * mode: 0 = 3d_iou (sum inter / sum union, the CLI's), 1 = avg_iou (mean of
 * the per-frame IoU over the union of frames, T/eval.py:99-117), 2 =
 * imagenetvid (fraction of frames with inter > 0.5 union, T/eval.py:51-70).
 *
 * taoamd_track_iou is the plan-less form: one lane per track pair walks the two
 * CSR lists with a two-pointer merge (slow; any input).
 *
 * taoamd_track_iou_planned is the fast form.  It follows a launch plan built
 * by taoamd_track_iou_plan_host, one workgroup per task:
 *   tasks      int32[n_tasks][4] {first row, rows (<= 32), first pair, pairs (<= 64)}
 *   task_rows  int32  the tasks' tracks: t for detection track t, n_dt + t for
 *              GT track t (= the row of trk_meta)
 *   task_pairs int32  detection row | GT row << 8 (rows local to the task)
 *   task_out   int64  index of the pair's result in `iou`
 * and reads the tracks' frames from the tasks' FRAME STREAM (taoamd_track_stream):
 *   frames     double[n_pieces * 8][4]  pieces of 8 consecutive timeline
 *              positions (one 256-byte piece = 8 boxes x, y, w, h; the far box
 *              where the track has no frame), in the order the tasks consume
 *              them: task by task, chunk by chunk of the task's timeline, the
 *              rows whose span reaches into the chunk in row order; piece 0 is
 *              eight far boxes
 *   task_base  int32[n_tasks]  a task's first piece
 * A task stages its tracks' frames in LDS chunk by chunk of the timeline and
 * adds the per-frame terms in ascending timeline order, one lane per pair.
 * Both forms give bit-identical results. */
int taoamd_track_iou(int64_t n_cells, const int32_t *cell_dt_off,
                     const int32_t *cell_gt_off, const int64_t *cell_iou_off,
                     int64_t n_pairs, const int32_t *dt_frame_off,
                     const int32_t *dt_frame_pos, const double *dt_frame_box,
                     const int32_t *gt_frame_off, const int32_t *gt_frame_pos,
                     const double *gt_frame_box, int32_t mode, double *iou,
                     int64_t *pair_frames, void *stream);
int taoamd_track_iou_planned(int64_t n_tasks, const int32_t *tasks,
                             const int32_t *task_rows, const int32_t *task_pairs,
                             const int64_t *task_out, const double *frames,
                             const int32_t *task_base, const int32_t *trk_meta,
                             int32_t mode, double *iou, int64_t *pair_frames,
                             void *stream);

/* The same IoUs when EVERY detection and ground-truth track has exactly one
 * frame (frame k of the lists = the frame of track k): a pair is one box IoU
 * if the two frames are the same timeline position, else 0 -- a single term,
 * so the value does not depend on any order of summation.  dt_group
 * (int32[n_dt][4], device) = {first GT track of the detection's cell, GT
 * tracks of the cell, the detection's place in its cell, the cell}. */
int taoamd_track_iou_single(int64_t n_dt, const int32_t *dt_group,
                            const int64_t *cell_iou_off, const int32_t *dt_frame_pos,
                            const double *dt_frame_box, const int32_t *gt_frame_pos,
                            const double *gt_frame_box, int32_t mode, double *iou,
                            int64_t *pair_frames, void *stream);

/* Padded frame table of a set of CSR tracks: track t owns the slots
 * meta[t].base_minus_first + p for p = first .. last (its first and last
 * timeline position); a slot holds the frame's box (x, y, w, h) or, where the
 * track has no frame, the far box (1e300, 1e300, 0, 0), against which every
 * intersection is exactly 0.  meta (int32[n_trk][4], device, filled by the
 * caller) = {first, last, base - first, 1 for a detection track / 0 for GT}; a
 * track without frames has first > last.  The call fills the slots
 * [slot_first, slot_first + n_slots) of `padded` (double[.][4]) with far boxes
 * and then scatters the n_frames frames.  One call per track set (detections,
 * ground truth) into disjoint slot ranges of ONE table whose slot 0 is a far
 * box (reserve it: slot_first = 0 for the first set, its tracks based at 1).
 * `inexact` (optional, int32[1], zeroed by the caller) is set when some box has a
 * coordinate that is not an integer below 2^20 -- with none the per-frame
 * products and their sums are exact and taoamd_track_iou_near has nothing to do. */
int taoamd_track_pad(int64_t n_trk, int64_t n_frames, const int32_t *frame_off,
                     const int32_t *frame_pos, const double *frame_box,
                     const int32_t *meta, int64_t slot_first, int64_t n_slots,
                     double *padded, int32_t *inexact, void *stream);

/* Frame stream of a launch plan (once per problem): the padded table's frames
 * rearranged into the order taoamd_track_iou_planned's tasks read them, so that
 * what a task requests for one chunk of its timeline is ONE contiguous stretch
 * of memory instead of 32 pieces scattered over the padded table.  A task's
 * chunks are the 8-position windows [8 c, 8 c + 8) from its earliest first to
 * its latest last position; row r of the task (rows in task_rows order, with
 * first <= last) owns a piece in every chunk its span first .. last reaches
 * into.  task_base[t] (int32, device, filled by the caller) = 1 + the pieces
 * of the tasks before t, where a task's pieces = the sum over its rows of
 * (last >> 3) - (first >> 3) + 1; `frames` holds 1 + the pieces of all tasks,
 * 256 bytes each, and a margin of 64 more pieces' bytes behind them (read, never
 * used, by taoamd_track_iou_planned).  Reads tasks, task_rows, trk_meta and the padded table
 * (which may be released afterwards). */
int taoamd_track_stream(int64_t n_tasks, const int32_t *tasks,
                        const int32_t *task_rows, const int32_t *trk_meta,
                        const int32_t *task_base, const double *padded,
                        double *frames, void *stream);

/* Guard of the documented frame-order deviation (frames are added in timeline
 * order; the reference adds them in CPython set order, T/eval.py:83-94): lists
 * the pairs whose IoU lies within max_ulp units in the last place of one of the
 * ten IoU thresholds or of another ground truth's IoU in the same row -- the
 * comparisons of the greedy match that a last-bit difference could flip.
 * *count = number of such pairs (zeroed by the call), list[0 .. min(count,
 * capacity)) = their indices into `iou`.  max_ulp must bound the reordering
 * error: adding n non-negative fp64 terms in two different orders moves a sum
 * by at most 2 (n - 1) units of roundoff, a quotient of two such sums by at
 * most 4 n - 2 units in the last place; callers pass 8 (n_dt + n_gt) + 8 for
 * the longest tracks (two rivals move independently).  The listed pairs are
 * recomputed in the reference's order by taoamd_track_iou_setorder. */
int taoamd_track_iou_near(int64_t n_cells, const int32_t *cell_gt_off,
                          const int64_t *cell_iou_off, int64_t n_pairs,
                          const double *iou, int32_t max_ulp, int32_t capacity,
                          int32_t *count, int64_t *list, void *stream);

/* The listed pairs recomputed on the device in the reference's order
 * (replaces compute_track_box_iou / compute_avg_track_iou, T/eval.py:73-117,
 * for exactly those pairs).  A thread per pair restates CPython's
 * ``set(gt_track.keys()) | set(dt_track.keys())`` (Objects/setobject.c of
 * 3.7 - 3.12, csrc/pyset.hpp), visits the union in slot order and adds the
 * per-frame terms of bb_intersect_union (T/eval.py:32-48) as the reference
 * does: mode 0 sums intersections and unions left to right, mode 1 takes
 * np.mean (numpy's pairwise summation) of the per-frame ratios.
 *   cell_unit      int32  video index of a cell
 *   tl_vid_start   int64  first timeline slot of a video in tl_image_id
 *   tl_image_id    int64  image id at (video, timeline position): the dict key
 *                         (all ids in [0, 2^61 - 1): hash(id) == id)
 *   count, list    device: the output of taoamd_track_iou_near (count read on
 *                  the device: no host round trip); `capacity` = entries of list
 *   scratch        int32[scratch_slots], 3 * table_cap slots per worker thread;
 *                  table_cap >= taoamd_track_iou_setorder_table(longest
 *                  detection track, longest GT track) in frames
 *   status         int32[1], bit 0 set if a pair needed a larger table (skipped)
 * Patches iou[list[k]] in place; asynchronous on `stream`. */
int64_t taoamd_track_iou_setorder_table(int64_t max_dt_frames, int64_t max_gt_frames);
int taoamd_track_iou_setorder(
    int64_t n_cells, const int32_t *cell_dt_off, const int32_t *cell_gt_off,
    const int64_t *cell_iou_off, const int32_t *cell_unit, const int64_t *tl_vid_start,
    const int64_t *tl_image_id, const int32_t *dt_frame_off, const int32_t *dt_frame_pos,
    const double *dt_frame_box, const int32_t *gt_frame_off, const int32_t *gt_frame_pos,
    const double *gt_frame_box, int32_t mode, const int32_t *count, int32_t capacity,
    const int64_t *list, double *iou, int32_t *scratch, int64_t scratch_slots,
    int64_t table_cap, int32_t *status, void *stream);

/* Host statements of the same text (test seams; synchronous, no GPU):
 * one pair's set-order IoU from host arrays (positions ascending, boxes x, y,
 * w, h; tl_image_id indexed by position), and the iteration order of
 * ``set(a) | set(b)`` for two lists of non-negative ints (out: n_a + n_b
 * entries). */
int taoamd_set_order_iou_host(const int64_t *tl_image_id, int32_t n_dt,
                              const int32_t *dt_pos, const double *dt_box, int32_t n_gt,
                              const int32_t *gt_pos, const double *gt_box, int32_t mode,
                              double *out);
int taoamd_pyset_union_order_host(int64_t n_a, const int64_t *a, int64_t n_b,
                                  const int64_t *b, int64_t *out, int64_t *n_out);

/* Launch plan of taoamd_track_iou_planned from HOST copies of the cell offsets
 * and of trk_meta (detection tracks first, then GT tracks).  Call with
 * tasks_host == NULL to get sizes[0..2] = tasks, task_rows entries, task_pairs
 * entries; then with buffers of 4 * sizes[0], sizes[1], sizes[2] int32 and
 * sizes[2] int64.  Every (detection track, GT track) pair of every cell lands
 * in exactly one task.  Synchronous, host only. */
int taoamd_track_iou_plan_host(int64_t n_cells, const int32_t *cell_dt_off_host,
                               const int32_t *cell_gt_off_host,
                               const int64_t *cell_iou_off_host,
                               const int32_t *trk_meta_host, int64_t *sizes,
                               int32_t *tasks_host, int32_t *task_rows_host,
                               int32_t *task_pairs_host, int64_t *task_out_host);

/* ---- greedy assignment --------------------------------------------------------
 * If dt_box/gt_box are non-NULL the IoU matrix of each cell is computed on the
 * fly from the boxes (LVIS, fused); otherwise it is read from `iou` at
 * cell_iou_off (TAO).  dst (optional, int32 per detection) redirects the
 * output row of detection d to dst[d] (used to write in (category, score)
 * order); NULL = identity.
 *   matched, ignored : uint64[n_dt * n_words]
 *   match_gt (opt.)  : int32[n_dt * n_rng*10], in-cell GT index or -1
 *                      (always in identity order)
 *   ious_out (opt.)  : LVIS only, double at cell_iou_off like `iou`
 * out_stride: distance, in 64-bit words, between consecutive output rows of
 * matched / ignored (0 = dense); lets the kernel write straight into an
 * interleaved exchange record.
 * Paired rows: when ignored == matched + 1 the two tables are read as ONE table
 * of (matched, ignored) pairs -- word w of row r is the pair at
 * matched[r * out_stride + 2 * w], dense out_stride = 2 * n_words -- and,
 * where matched is 16-byte aligned and out_stride even, a pair is written with
 * one 16-byte store;
 * taoamd_accumulate* recognise the same layout by the same pointer relation.
 * The pointer relation IS the layout flag of this ABI: `ignored == matched + 1`
 * means interleaved pairs in taoamd_match and every taoamd_accumulate* entry;
 * the entry that only knows DENSE [n][n_words] tables -- taoamd_gather_rows
 * (its destination) -- returns TAOAMD_ERR_ARG when
 * handed that relation instead of scrambling rows (tests/test_abi.py).
 * max_gt_per_cell must be >= the largest GT count of a cell (host knows it
 * from the CSR table); cells with more than 64 GTs take a slower kernel and
 * more than TAOAMD_MAX_GT_PER_CELL is an error.
 * Launch plan (optional, all NULL/0 = one wavefront per cell): `groups`
 * (int32[n_groups][4], device) lists runs of consecutive cells with at most
 * 64 detections and 64 GTs in total and at most 8 GTs per cell, as {first
 * detection, detections, first GT, GTs} of the run -- one wavefront evaluates
 * a whole run with coalesced loads; `singles` (int32[n_singles]) lists the
 * remaining cells that hold detections; `dt_group` (int32[n_dt][4]) gives per
 * detection {first GT of its cell, GT count of its cell, its position inside
 * the cell, cell index}, so the kernel reaches everything with two dependent
 * loads instead of walking detection -> cell -> cell tables.  `dt_meta`
 * (optional, uint32[n_dt]; used where the kernel computes the IoUs itself and
 * stores none) packs what a run's wavefront needs of that row and the flag
 * byte into one word: dt_flags | (first GT of the cell - first GT of the run)
 * << 8 | (GT count of the cell) << 14 | (position inside the cell) << 18 --
 * 4 instead of 17 bytes read per detection. */
#define TAOAMD_MAX_GT_PER_CELL 3072
int taoamd_match(int64_t n_cells, const int32_t *cell_dt_off,
                 const int32_t *cell_gt_off, const int64_t *cell_iou_off,
                 int32_t max_gt_per_cell, const double *dt_box,
                 const double *gt_box, const double *iou, int32_t n_rng,
                 const uint32_t *gt_rng, const uint32_t *dt_rng,
                 const uint8_t *gt_flags, const uint8_t *dt_flags,
                 const int32_t *dst, int64_t out_stride, uint64_t *matched,
                 uint64_t *ignored, int32_t *match_gt, double *ious_out,
                 const int32_t *dt_group, const uint32_t *dt_meta,
                 const int32_t *groups, int32_t n_groups, const int32_t *singles,
                 int32_t n_singles, void *stream);

/* ---- cell-table build, detection side (csrc/flatten.hip) ----------------------
 * The per-box half of what the reference does with dicts in L/results.py:20-84,
 * L/lvis.py:90-96, L/eval.py:59-110 (and the T/ counterparts): the host keeps
 * the ground-truth half and everything that is O(cells) (flatten_dev.py).
 *
 * taoamd_flat_map     image / category ids -> their positions in the sorted
 *                     unique id lists of the ground truth (-1 = unknown),
 *                     area = w * h unless `area_in` is given,
 *                     boxes per image (img_count[n_img + 1]) and their
 *                     exclusive scan (img_start[n_img + 1]), optionally the
 *                     first box of every image (img_first[n_img], file order;
 *                     0x7f7f7f7f = none); status[0] = boxes of unknown images,
 *                     status[1] = most boxes in one image
 * taoamd_flat_rank_drop   dropped[d] = 1 for boxes beyond the best max_dets of
 *                     their image; `order` = the stable sort by (image, -score)
 * taoamd_flat_filter  key[d] = cat * n_unit + unit for boxes that survive the
 *                     known-category, 0 < area < inf and federated filters
 *                     (the cell has ground truth: key in sorted gkeys; or the
 *                     category is in the unit's negative list), INT32_MAX for
 *                     the others; flags[d] = DT_IGNORE_UNMATCHED when the
 *                     category is in the unit's not-exhaustive list (or, with
 *                     area_flag, the area is outside [0, 1e10]); *n_keep =
 *                     survivors.  unit_row[unit] = row of the CSR lists
 * taoamd_flat_gather  rows of the first n_keep entries of `order` (the stable
 *                     sort by (key, -score)): source row, score, flags, key,
 *                     category = key / n_unit, box (optional)
 * taoamd_flat_runs    runs of equal values of a sorted key array: run_id[i],
 *                     run_key[r], run_start[r], *n_runs
 * taoamd_flat_remap   out[i] = map[id[i]] */
int taoamd_flat_map(int64_t n, const int64_t *image_id, const int64_t *category_id,
                    const double *bbox, const double *area_in, int64_t n_img,
                    const int64_t *img_ids, int64_t n_cat, const int64_t *cat_ids,
                    int32_t *img, int32_t *cat, double *area, int32_t *img_count,
                    int32_t *img_start, int32_t *img_first, int32_t *status,
                    void *stream);
int taoamd_flat_rank_drop(int64_t n, const int32_t *order, const int32_t *img,
                          const int32_t *img_start, int32_t max_dets,
                          uint8_t *dropped, void *stream);
int taoamd_flat_filter(int64_t n, const int32_t *unit, const int32_t *cat,
                       const double *area, const int64_t *category_id,
                       const uint8_t *dropped, int32_t n_unit, int64_t n_gkeys,
                       const int32_t *gkeys, const int32_t *unit_row,
                       const int64_t *neg_off, const int64_t *neg_val,
                       const int64_t *nel_off, const int64_t *nel_val,
                       int32_t area_flag, int32_t *key, uint8_t *flags,
                       int32_t *n_keep, void *stream);
int taoamd_flat_gather(int64_t n_keep, const int32_t *order, const double *score,
                       const uint8_t *flags, const int32_t *key,
                       const double *bbox, int32_t n_unit, int32_t *dt_row,
                       double *dt_score, uint8_t *dt_flags, int32_t *dt_key,
                       int32_t *dt_cat, double *dt_box, void *stream);
/* Track level (T/results.py:20-132, T/tao.py:108-254, T/eval.py:196-243); see
 * csrc/flatten.hip for the stages and flatten_dev.flatten_tao_device for the
 * order they run in.  Sort keys travel as exact doubles, negated (the radix
 * sort is descending in its score argument).
 *   taoamd_flat_ordscore   score of boxes in images beyond max_dets, else 0
 *   taoamd_flat_ordinal    ordinal[d] = place of box d inside its image in the
 *                          post-truncation list, dropped[d] = beyond max_dets
 *   taoamd_flat_merge_cat  merged category id + its index (-1 unknown)
 *   taoamd_flat_split64 / _compose / _gather_cols   helpers of the sorts
 *   taoamd_flat_track_of   trk[d] = run of its track id; status[2] = a box whose
 *                          track spans two videos (else unchanged)
 *   taoamd_flat_keys       list-order / visiting-order / frame / timeline keys,
 *                          track keys of the kept and of the selected boxes
 *   taoamd_flat_track_kept per track: score (np.mean when the boxes' scores
 *                          differ: status[1] = 1), first kept box; status[3] =
 *                          a box whose category differs inside its track
 *   taoamd_flat_track_sel  per track: mean area (left to right), boxes,
 *                          distinct images, first appearance
 *   taoamd_flat_track_filter  federated filter on the video lists, flags, key
 *   taoamd_flat_frames     frame lists of the final tracks
 *   taoamd_flat_scan       exclusive scan of int32 counts (start[n] = total) */
/* *count = kept boxes with a negative corner or an empty side (T/results.py:100-103) */
int taoamd_flat_count_bad(int64_t n, const double *bbox, const uint8_t *dropped,
                          int32_t *count, void *stream);
int taoamd_flat_ordscore(int64_t n, const int32_t *img, const int32_t *img_count,
                         const double *score, int32_t max_dets, double *out,
                         void *stream);
int taoamd_flat_ordinal(int64_t n, const int32_t *order, const int32_t *img,
                        const int32_t *img_start, int32_t max_dets,
                        int32_t *ordinal, uint8_t *dropped, void *stream);
int taoamd_flat_merge_cat(int64_t n, const int64_t *category_id, int64_t n_merge,
                          const int64_t *merge_src, const int64_t *merge_dst,
                          int64_t n_cat, const int64_t *cat_ids, int64_t *merged_id,
                          int32_t *cat, void *stream);
int taoamd_flat_split64(int64_t n, const int64_t *key, const int32_t *order,
                        int32_t *lo, int32_t *hi, void *stream);
int taoamd_flat_compose(int64_t n, const int32_t *outer, const int32_t *inner,
                        int32_t *out, void *stream);
int taoamd_flat_gather_cols(int64_t n, const int32_t *idx, const int32_t *src_i32,
                            int32_t *out_i32, const double *src_f64,
                            double *out_f64, const uint8_t *src_u8,
                            uint8_t *out_u8, void *stream);
int taoamd_flat_track_of(int64_t n, const int32_t *order, const int32_t *run_id,
                         const int32_t *run_start, const int64_t *video_id,
                         int32_t *trk, int32_t *status, void *stream);
int taoamd_flat_keys(int64_t n, const int32_t *img, const int32_t *ordinal,
                     const uint8_t *dropped, const int32_t *cat, const double *area,
                     const int32_t *trk, const int32_t *img_rank,
                     const int32_t *visit_rank, const double *img_frame,
                     const int32_t *tl_pos, double M, double *keep_key,
                     double *visit_key, double *frame_key, double *pos_key,
                     int32_t *trk_keep, int32_t *trk_sel, void *stream);
int taoamd_flat_track_kept(int64_t n_runs, const int32_t *run_trk,
                           const int32_t *run_start, int64_t n_sorted,
                           const int32_t *order_k, const double *score,
                           const int64_t *merged_id, double *trk_score,
                           int32_t *trk_first_kept, int32_t *status, void *stream);
int taoamd_flat_track_sel(int64_t n_runs, const int32_t *run_trk,
                          const int32_t *run_start, int64_t n_sorted,
                          const int32_t *order_s, const double *area,
                          const double *visit_key, const int32_t *img,
                          double *sel_area, int32_t *sel_len, int32_t *sel_frames,
                          double *sel_first, void *stream);
int taoamd_flat_track_filter(int64_t n_runs, const int32_t *run_trk,
                             const int32_t *sel_len, const int32_t *trk_first_kept,
                             const int32_t *cat, const int64_t *merged_id,
                             const int64_t *video_id, const int64_t *track_id,
                             int64_t n_vid, const int64_t *vid_ids, int64_t n_gkeys,
                             const int32_t *gkeys, const int32_t *vid_row,
                             const int64_t *neg_off, const int64_t *neg_val,
                             const int64_t *nel_off, const int64_t *nel_val,
                             int32_t *key, uint8_t *flags, int32_t *n_keep,
                             int32_t *status, void *stream);
int taoamd_flat_frames(int64_t n_trk, const int32_t *trk_run,
                       const int32_t *run_start, int64_t n_runs, int64_t n_sorted,
                       const int32_t *order_s, const int32_t *img,
                       const int32_t *tl_pos, const double *bbox,
                       const int32_t *frame_off, int32_t *frame_pos,
                       double *frame_box, void *stream);
int taoamd_flat_scan(int64_t n, const int32_t *count, int32_t *start,
                     int32_t *status2, void *stream);
size_t taoamd_flat_runs_workspace(int64_t n);
int taoamd_flat_runs(int64_t n, const int32_t *sorted_key, int32_t *run_id,
                     int32_t *run_key, int32_t *run_start, int32_t *n_runs,
                     void *workspace, size_t workspace_bytes, void *stream);
int taoamd_flat_runs_by(int64_t n, const int32_t *key, const int32_t *order,
                        int32_t *run_id, int32_t *run_key, int32_t *run_start,
                        int32_t *n_runs, void *workspace, size_t workspace_bytes,
                        void *stream);
int taoamd_flat_runs64_by(int64_t n, const int64_t *key, const int32_t *order,
                          int32_t *run_id, int64_t *run_key, int32_t *run_start,
                          int32_t *n_runs, void *workspace, size_t workspace_bytes,
                          void *stream);
int taoamd_flat_remap(int64_t n, const int32_t *id, const int32_t *map,
                      int32_t *out, void *stream);

/* Launch plan of taoamd_match (groups / singles arguments) from HOST copies of
 * the cell offsets: runs of consecutive small cells (each <= cap_d detections
 * and <= cap_cell_g ground truths, a run <= cap_d detections and <= cap_g
 * ground truths) -> groups[.][2] = {first cell, end cell}; other cells with
 * detections -> singles.  groups == NULL: sizes only (sizes[0] groups,
 * sizes[1] singles).  The kernels' caps are 64, 64, 8.  Synchronous, host. */
int taoamd_match_plan_host(int64_t n_cells, const int32_t *cell_dt_off_host,
                           const int32_t *cell_gt_off_host, int32_t cap_d,
                           int32_t cap_g, int32_t cap_cell_g, int64_t *sizes,
                           int32_t *groups_host, int32_t *singles_host);

/* ---- stable sort by (category asc, score desc) ---------------------------------
 * order[p] = detection at sorted position p; dst[d] = sorted position of
 * detection d (either may be NULL).  Ties keep input order, i.e. the
 * reference's concatenation order.  dt_cat: any non-negative int32 key;
 * dt_score may be NULL (integer key alone).  Workspace:
 * taoamd_sort_workspace(n). */
size_t taoamd_sort_workspace(int64_t n);
int taoamd_sort_by_cat_score(int64_t n, const int32_t *dt_cat,
                             const double *dt_score, int32_t *order,
                             int32_t *dst, void *workspace,
                             size_t workspace_bytes, void *stream);

/* The same order by SAMPLE SORT (round 3; what the evaluator passes call):
 * every category -- or chunk of TAOAMD_SORT_CHUNK elements of a longer one --
 * is cut into buckets of ~416 elements by splitters taken from a sorted sample,
 * one pass scatters the elements into their buckets, one wavefront sorts a
 * bucket in registers (bitonic network over <= 1024 (key, index) pairs) and
 * writes order / dst; categories longer than a chunk are finished by the
 * merge-path passes of taoamd_sort_segments.  Replaces np.argsort(-dt_scores,
 * kind="mergesort") of lvis_amodal/eval.py:353-361 / tao_amodal/eval.py:508-518.
 * The plan depends on cat_off alone: taoamd_sort_plan_host (host, synchronous;
 * call with chunks == NULL for sizes[0..4] = chunks, split chunks, scatter
 * tiles, buckets, 1 if some category is longer than a chunk; then with buffers
 * of 8 * sizes[0], sizes[1], sizes[2], sizes[3] int32) -- the tables are
 * uploaded by the caller and stay valid while cat_off does.  tile_off / n_tiles
 * as for taoamd_sort_segments (used by the merge passes only).
 * taoamd_sort_sampled_cap_limit: test seam -- buckets beyond `limit` elements
 * take the overflow path (ranking by counting); 0 restores the default. */
#define TAOAMD_SORT_CHUNK (16 * 2816)
int taoamd_sort_plan_host(int32_t n_cat, const int32_t *cat_off_host, int64_t *sizes,
                          int32_t *chunks, int32_t *split_list, int32_t *stile_chunk,
                          int32_t *bucket_chunk);
size_t taoamd_sort_sampled_workspace(int64_t n, int64_t n_buckets, int32_t merge);
int taoamd_sort_sampled_cap_limit(int32_t limit);
/* Ordering other streams around the splitter kernel INSIDE taoamd_sort_sampled:
 * `event` (from taoamd_event_create) is recorded on the sort's stream right
 * behind the splitter kernel of the calling thread's NEXT taoamd_sort_sampled
 * (NULL = none; the request is consumed by that call).  The image level's
 * splitters are 0.06 ms of latency at the head of the step's critical chain;
 * work started beside the sort (the track level's 3D IoU) waits for them with
 * taoamd_stream_wait_event instead of starving their workgroups of wave slots. */
int taoamd_sort_sampled_notify(void *event);
int taoamd_event_create(void **event);
int taoamd_event_destroy(void *event);
int taoamd_stream_wait_event(void *stream, void *event);
int taoamd_sort_sampled(int64_t n, int32_t n_cat, const int32_t *cat_off,
                        const int32_t *tile_off, int32_t n_tiles, int32_t max_segment,
                        const double *dt_score, int32_t n_chunks, const int32_t *chunks,
                        int32_t n_split, const int32_t *split_list, int32_t n_stiles,
                        const int32_t *stile_chunk, int32_t n_buckets,
                        const int32_t *bucket_chunk, int32_t *order, int32_t *dst,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Same result for detections that are already grouped by category (the
 * category-major cell tables).  cat_off (int32[n_cat+1], device) delimits the
 * runs; every run is cut into tiles of TAOAMD_SEGMENT_TILE elements,
 * tile_off (int32[n_cat+1], device) = exclusive prefix of ceil(len/TILE),
 * n_tiles its total; max_segment = host value of the longest run.  Tiles are
 * sorted in LDS, longer runs finished by rank-merge passes.
 * Workspace: taoamd_sort_segments_workspace(n). */
#define TAOAMD_SEGMENT_TILE 2816
size_t taoamd_sort_segments_workspace(int64_t n);
int taoamd_sort_segments(int64_t n, int32_t n_cat, const int32_t *cat_off,
                         const int32_t *tile_off, int32_t n_tiles,
                         int32_t max_segment, const int32_t *dt_cat,
                         const double *dt_score, int32_t *order, int32_t *dst,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- row gather -----------------------------------------------------------------
 * dst_*[p*n_words + w] = src_*[order[p]*src_stride + w]: brings exchanged
 * records (multi-GPU path) into sorted order.  Destination tables dense (the
 * pair layout of taoamd_match is refused: TAOAMD_ERR_ARG). */
int taoamd_gather_rows(int64_t n, int32_t n_words, const uint64_t *src_matched,
                       const uint64_t *src_ignored, int64_t src_stride,
                       const int32_t *order, uint64_t *dst_matched,
                       uint64_t *dst_ignored, void *stream);

/* ---- accumulate -------------------------------------------------------------------
 * matched/ignored are in sorted order (row p = sorted position p); cat_off
 * (int32[n_cat+1], device) delimits the categories in that order.  Outputs, C
 * order, -1 where the category has no evaluated GT in the range:
 *   precision[T][R][n_cat][n_rng], recall[T][n_cat][n_rng]
 * Workspace: taoamd_accumulate_workspace(n_dt, n_cat, n_rng). */
size_t taoamd_accumulate_workspace(int64_t n_dt, int32_t n_cat, int32_t n_rng);
/* The two halves of taoamd_accumulate, used separately by the multi-GPU path:
 *  _compact  sweeps the categories [k_begin, k_end) (rows of other categories
 *            must be absent) and writes the category-major tables
 *              val[n_cat][n_rng][T][R]   precision at the recall thresholds,
 *                                        as opaque 8-byte records (tp << 32 |
 *                                        tp + fp of the row that sets the
 *                                        value; declared double for size)
 *              rec[n_cat][n_rng][T]      recall
 *            for every (k, range) with num_gt > 0 in that category range;
 *  _finalize turns complete tables into the reference layout (-1 fill) and a
 *            record into tp / (fp + tp + eps), L/eval.py:384.
 * Category-major tables make a rank's share one contiguous block, so ranks
 * exchange them with a single all-gather.
 * max_segment: rows of the longest category among those swept (the host
 * knows it from cat_off), or 0 if unknown.  When every category fits one
 * workgroup (<= 4096 rows with one combo word, <= 1024 with four) the sweep
 * is a single fused launch with all intermediates in LDS / registers;
 * otherwise (or with 0) categories are cut into chunks spread over the chip.
 * Workspace of _compact: taoamd_accumulate_workspace (val/rec excluded). */
size_t taoamd_compact_elems(int32_t n_cat, int32_t n_rng); /* 8-byte elements in val */
int taoamd_accumulate_compact(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                              const int32_t *cat_off, const uint64_t *matched,
                              const uint64_t *ignored, const int32_t *num_gt,
                              int32_t k_begin, int32_t k_end, int32_t max_segment,
                              double *val, double *rec, void *workspace,
                              size_t workspace_bytes, void *stream);
int taoamd_finalize(int32_t n_cat, int32_t n_rng, const int32_t *num_gt,
                    const double *val, const double *rec, double *precision,
                    double *recall, void *stream);
int taoamd_accumulate(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                      const int32_t *cat_off, const uint64_t *matched,
                      const uint64_t *ignored,
                      const int32_t *num_gt, int32_t max_segment, double *precision,
                      double *recall, void *workspace, size_t workspace_bytes,
                      void *stream);
/* The same in two steps, for a caller that sweeps the same categories again and
 * again (an evaluation plan replayed per pass): _prepare builds the table that
 * cuts the categories into chunks -- it depends on cat_off alone -- in the
 * workspace, _prepared is taoamd_accumulate without that launch.  Same
 * workspace, n_dt, n_cat, n_rng, cat_off and max_segment in both; nothing else
 * may use the workspace in between. */
/* The one-pass sweep of long categories resolves the counts of a category's
 * earlier rows by a decoupled look-back between workgroups; a wait that does
 * not end within the poll limit (never observed: it would mean workgroups do
 * not start in order, which an idle GPU does and a shared one need not) gives
 * up and sets a flag in the workspace instead of hanging the GPU.
 * *flag_host != 0: the tables of a pass on this workspace are not to be
 * trusted -- run the pass's sweep again with taoamd_accumulate_chunked /
 * taoamd_accumulate_compact_chunked (same arguments as taoamd_accumulate /
 * _compact; the chunked kernels whatever the mode; the rows of the pass are
 * still in place; a prepared plan in the workspace does not survive it, so
 * taoamd_accumulate_prepare again).  Synchronises with `stream`. */
int taoamd_accumulate_error(const void *workspace, void *stream, int32_t *flag_host);
int taoamd_accumulate_chunked(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                      const int32_t *cat_off, const uint64_t *matched,
                      const uint64_t *ignored,
                      const int32_t *num_gt, int32_t max_segment, double *precision,
                      double *recall, void *workspace, size_t workspace_bytes,
                      void *stream);
int taoamd_accumulate_compact_chunked(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                              const int32_t *cat_off, const uint64_t *matched,
                              const uint64_t *ignored, const int32_t *num_gt,
                              int32_t k_begin, int32_t k_end, int32_t max_segment,
                              double *val, double *rec, void *workspace,
                              size_t workspace_bytes, void *stream);
/* Which sweep long categories take, for the process: -1 automatic (the
 * environment's TAOAMD_SWEEP=chunked|lookback|twopass if set, else the
 * one-pass look-back sweep from 6 M rows up), 0 the chunked kernels, 1 the
 * one-pass sweep with the look-back, 2 the one-pass sweep behind a counting
 * pass.  An explicit 1 / 2 also takes the short categories the fused
 * single-workgroup sweep would have taken (every case of the parity suite runs
 * under each mode that way).  _plan_kind: the kind of plan
 * taoamd_accumulate_prepare builds for these sizes under the current mode (0
 * none, 1 chunk table, 2 / 3 super-chunk table): a prepared workspace serves
 * taoamd_accumulate_prepared while this is the value it was built under.
 * _spin_limit: polls a look-back waits for one predecessor (0: the default,
 * 2^18; < 0: every look-back gives up at once -- fault injection for the
 * caller's recovery path).
 * _giveup_counter: a caller-owned DEVICE word (NULL: none) to which every
 * look-back that gives up adds one, in every pass launched while it is
 * registered, and which the library never clears -- the workspace's flag is per
 * pass (an unprepared pass zeroes it when it starts), so a caller that runs
 * many passes and synchronises once reads this word instead.
 * max_segment, wherever an entry point of this family takes it: 0 = unknown;
 * a positive value MUST be an upper bound of the longest category's rows -- it
 * chooses the single-workgroup kernels and sizes the one-pass sweep's XCD-aware
 * launch, and rows of a category longer than stated would not be swept. */
int taoamd_accumulate_sweep_mode(int32_t mode);
int taoamd_accumulate_plan_kind(int64_t n_dt, int32_t n_rng, int32_t max_segment);
int taoamd_accumulate_spin_limit(int32_t polls);
int taoamd_accumulate_giveup_counter(uint32_t *device_word);
int taoamd_accumulate_prepare(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                              const int32_t *cat_off, int32_t max_segment,
                              void *workspace, size_t workspace_bytes, void *stream);
int taoamd_accumulate_prepared(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                      const int32_t *cat_off, const uint64_t *matched,
                      const uint64_t *ignored,
                      const int32_t *num_gt, int32_t max_segment, double *precision,
                      double *recall, void *workspace, size_t workspace_bytes,
                      void *stream);
/* The same with the rows where the match kernel left them when it ran WITHOUT
 * `dst` (row = detection, cell order): order[p] = detection at sorted position
 * p (taoamd_sort_segments).  The first sweep gathers the rows through it -- the
 * match then neither waits for the sort nor scatters 16-byte rows. */
int taoamd_accumulate_by_order(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                               const int32_t *cat_off, const int32_t *order,
                               const uint64_t *matched, const uint64_t *ignored,
                               const int32_t *num_gt, int32_t max_segment,
                               double *precision, double *recall, void *workspace,
                               size_t workspace_bytes, void *stream);

/* ---- multi-GPU result exchange (category-partitioned evaluation) -------------------
 * No reference counterpart (the reference is single-process); these carry the
 * tables of taoamd_accumulate_compact between ranks and end in the layout of
 * taoamd_finalize, i.e. of `eval["precision"]` / `eval["recall"]`
 * (lvis_amodal/eval.py:366-371,420-426, tao_amodal/eval.py:521-526,576-584).
 * Rank b owns the categories [b * block_cats, (b+1) * block_cats); a row is a
 * (category, range) pair, global row = category * n_rng + range.
 *
 * A chunk holds one rank's share: the num_gt, level offset and rec of its rows and, per row
 * with num_gt > 0 and per IoU threshold, the distinct runs ("levels") of the
 * 101 recall columns -- a row with n ground truths has at most min(n,100)+1
 * of them, and which columns coincide follows from n alone, so receivers
 * rebuild the map from the chunk header.  Chunks of all ranks have
 * taoamd_exchange_chunk_bytes(block_cats, n_rng, capacity) bytes and sit
 * back to back in `chunks` (one in-place all-gather moves them).
 *  _sizes   levels needed by every block (device int64[world]) from the
 *           gathered num_gt[world * block_cats][n_rng]; capacity = their max
 *  _pack    own rows of val / rec / num_gt (tables addressed by GLOBAL row,
 *           as written by taoamd_accumulate_compact) -> `chunk` (this rank's)
 *  _unpack  all chunks -> precision[T][R][n_cat][n_rng], recall[T][n_cat][n_rng]
 *           (-1 fill) and optionally the assembled num_gt[n_cat][n_rng]
 * *overflow (device int32, may be NULL) is OR-ed with 1 if a block needs more
 * than `capacity` levels (the excess is dropped, results are then invalid).
 * maps_ready != 0: the run maps and level offsets of ALL rows are already in
 * the workspace -- a _sizes call on the num_gt of all rows earlier in the
 * pass (a rank of the by-video partition knows them from its all-reduce, long
 * before the sweep ends) -- so _pack / _unpack start their copy at once.
 * Workspace: taoamd_exchange_workspace(block_cats, n_rng, world). */
size_t taoamd_exchange_chunk_bytes(int32_t block_cats, int32_t n_rng, int64_t capacity);
size_t taoamd_exchange_workspace(int32_t block_cats, int32_t n_rng, int32_t world);
int taoamd_exchange_sizes(int32_t block_cats, int32_t n_rng, int32_t world,
                          const int32_t *num_gt, int64_t *totals, void *workspace,
                          size_t workspace_bytes, void *stream);
int taoamd_exchange_pack(int32_t n_cat, int32_t n_rng, int32_t block_cats,
                         int32_t world, int32_t rank, const int32_t *num_gt,
                         const double *val, const double *rec, void *chunk,
                         int64_t capacity, int32_t *overflow, void *workspace,
                         size_t workspace_bytes, int32_t maps_ready, void *stream);
int taoamd_exchange_unpack(int32_t n_cat, int32_t n_rng, int32_t block_cats,
                           int32_t world, const void *chunks, int64_t capacity,
                           int32_t *num_gt_out, double *precision, double *recall,
                           int32_t *overflow, void *workspace,
                           size_t workspace_bytes, int32_t maps_ready, void *stream);

/* By-video partition, owner side (round 5: in two messages).  A rank's records
 * lie at their sorted place (category, -score): the sort gives the place, the
 * match writes a detection's (matched, ignored) pairs there.  The merged input
 * of an owner is, for source s, the rows [src_base[s], src_base[s + 1]) -- the
 * wire buffer = what the all_to_all delivered, the sources' rows back to back
 * WITHOUT those of `own_rank` (>= 0), which never travel: they are read where
 * the rank's own sort / match wrote them (own_rank < 0: every source's rows
 * are in the wire buffer).  A record carries no category, the run it lies in
 * says it: run_off[s * (block_cats + 1) + kb] = offset of the run of the
 * block's category kb inside source s's rows; cat_base[kb] = first row of that
 * category in the merged layout.
 *  _scores     out[dst[i]] = score[i] (bit patterns): the first message, ready
 *              when the local sort is;
 *  _positions  pos[i] = row of record i in the reference's order (stable -score
 *              sort of the sources' concatenation in rank order,
 *              L/eval.py:353-361): one binary search per other source over the
 *              exchanged scores instead of a radix sort -- needs the scores
 *              alone, so it runs beside the match;
 *  _place      the second message's 16-byte (matched, ignored) pairs scattered
 *              to out[pos[i] * n_words + w]: the paired table the sweep reads
 *              (out, out + 1 as taoamd_accumulate*'s matched / ignored). */
int taoamd_exchange_scores(int64_t n, const int32_t *dst, const double *score,
                           int64_t *out, void *stream);
int taoamd_exchange_positions(int64_t n_recv, int32_t world, int32_t block_cats,
                              const int64_t *scores, const int64_t *own_scores,
                              int32_t own_rank, const int64_t *src_base,
                              const int64_t *run_off, const int64_t *cat_base,
                              int32_t *pos, void *stream);
int taoamd_exchange_place(int64_t n_recv, int32_t world, int32_t n_words,
                          const uint64_t *rows, const uint64_t *own_rows,
                          int32_t own_rank, const int64_t *src_base,
                          const int32_t *pos, uint64_t *out, void *stream);

/* ---------------------------------------------------------------------------
 * The prediction file read on the device (csrc/json_ingest.hip): replaces the
 * reference's json.load of prediction.json (lvis_amodal/results.py:29-30,
 * tools/eval_on_tao_amodal.py:127-128) for the six keys of a prediction.
 *
 * taoamd_json_pred_open copies the file's text into HBM, finds the list's
 * objects (string state, bracket depth and object count of 16 KB blocks, three
 * prefix sums) and checks the list's shape.  *status: TAOAMD_OK (a handle is
 * returned); TAOAMD_JSON_FALLBACK -- the file holds something this reader leaves
 * to the host's (a backslash, an element that is not an object, text around the
 * list, an empty file): call the host reader (include/tao_amodal_ingest.h),
 * whose results and error messages are the contract; TAOAMD_ERR_ARG -- the file
 * cannot be opened (err says so); TAOAMD_ERR_HIP.  `work` (optional, device
 * memory of taoamd_json_pred_workspace(file size) bytes, the caller's until
 * taoamd_json_pred_close): the text, the tables and the objects' offsets live
 * there instead of in memory the call allocates -- for callers with a pool.
 *
 * taoamd_json_pred_convert converts the objects (one thread each; decimal ->
 * double correctly rounded, csrc/decfloat.hpp) into the caller's DEVICE arrays of
 * taoamd_json_pred_count() rows (bbox: 4 doubles a row) and returns when they
 * are written; taoamd_json_pred_read does the same into HOST arrays.
 * Objects left to the host reader -- literals, numbers of more than 19
 * digits, ids that are not plain integers, missing keys, unexpected syntax --
 * are listed (HOST arrays): flag[k] = the object's number, flag_at[k] = the byte
 * offset of its '{' in the file, for the first flag_cap of *n_flagged; their
 * rows are not written (taoamd_pred_patch of the host library fills them in
 * host arrays or reports the error).  More than flag_cap: use the host reader
 * for the file. */
size_t taoamd_json_pred_workspace(size_t file_bytes);
void *taoamd_json_pred_open(const char *path, void *work, size_t work_bytes,
                            int32_t *status, char *err, size_t errlen, void *stream);
int64_t taoamd_json_pred_count(void *handle);
int taoamd_json_pred_convert(void *handle, int64_t *image_id, int64_t *category_id,
                             double *bbox, double *score, int64_t *track_id,
                             int64_t *video_id, int64_t *flag, int64_t *flag_at,
                             int32_t flag_cap, int32_t *n_flagged);
int taoamd_json_pred_read(void *handle, int64_t *image_id, int64_t *category_id,
                          double *bbox, double *score, int64_t *track_id,
                          int64_t *video_id, int64_t *flag, int64_t *flag_at,
                          int32_t flag_cap, int32_t *n_flagged);
void taoamd_json_pred_close(void *handle);

#ifdef __cplusplus
}
#endif
#endif
