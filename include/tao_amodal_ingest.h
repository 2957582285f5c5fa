/* tao_amodal_ingest.h -- C ABI of the native columnar readers (host only).
 *
 * libtao_amodal_ingest.so (tao_amodal_amd/csrc/ingest.cpp) turns the two JSON
 * inputs of the evaluation path into the contiguous columns the cell tables
 * are built from.  What it replaces in the reference:
 *
 *   prediction list   json.load + list-of-dict handling
 *                       tools/eval_on_tao_amodal.py:127-128,
 *                       tao_amodal/evaluation/lvis_amodal/results.py:29-30,
 *                       tao_amodal/evaluation/tao_amodal/results.py:33-35
 *   annotation file   json.load + _create_index
 *                       tao_amodal/evaluation/lvis_amodal/lvis.py:34-61,
 *                       tao_amodal/evaluation/tao_amodal/tao.py:84-160
 *
 * Numbers are converted with std::from_chars, i.e. to the value json.load
 * gives; integer ids keep 64 bits.  Every function is thread safe; a parse
 * uses all OpenMP threads.  On error the parse functions return NULL and put
 * a message into err: "cannot open <path>", "KeyError: '<key>'" when a key
 * the reference indexes unconditionally is absent, otherwise a description
 * of the malformed spot.  Handles are freed by the caller.
 */
#ifndef TAO_AMODAL_INGEST_H
#define TAO_AMODAL_INGEST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- prediction list: [{image_id, category_id, bbox[4], score, track_id?, video_id?}, ...]
 * Missing track_id / video_id become -1 (results.py's r.get(...)).
 * "results is not a list." when the document is not a list. */
void *taoamd_pred_parse(const char *path, char *err, size_t errlen);
/* One rank's share of the list in a multi-process run (torchrun
 * tools/eval_on_tao_amodal.py): every process scans the file's structure, the
 * elements [n * part / n_parts, n * (part + 1) / n_parts) are converted;
 * taoamd_pred_part_info gives the share's first position and n. */
void *taoamd_pred_parse_part(const char *path, int64_t part, int64_t n_parts, char *err,
                             size_t errlen);
void taoamd_pred_part_info(void *handle, int64_t *first, int64_t *total);
/* The same in two steps, converting straight into caller memory (what
 * DTColumns.from_file_native does: numpy arrays, first touched by all cores):
 * scan -> info (first position, records of the share, records of the file) ->
 * convert into n int64 / 4n double / n double / n int64 / n int64 arrays
 * (0 = ok, 2 = malformed record, message in err) -> free. */
void *taoamd_pred_scan(const char *path, int64_t part, int64_t n_parts, char *err,
                       size_t errlen);
void taoamd_pred_scan_info(void *handle, int64_t *first, int64_t *count, int64_t *total);
int taoamd_pred_convert(void *handle, int64_t *image_id, int64_t *category_id,
                        double *bbox, double *score, int64_t *track_id,
                        int64_t *video_id, char *err, size_t errlen);
void taoamd_pred_scan_free(void *handle);
/* Rows the device-side reader (taoamd_json_pred_read, include/tao_amodal_hip.h)
 * left to this one: object idx[k] of the file's list, whose '{' is byte at[k]
 * of the file, is read into row idx[k] of the columns.  0 = ok, 1 = the file
 * cannot be read, 2 = malformed record (message of the first one in err). */
int taoamd_pred_patch(const char *path, int64_t n, const int64_t *idx, const int64_t *at,
                      int64_t *image_id, int64_t *category_id, double *bbox,
                      double *score, int64_t *track_id, int64_t *video_id, char *err,
                      size_t errlen);
int64_t taoamd_pred_count(void *handle);
/* copies the columns into caller memory: n int64 / 4n double / n double / ... */
int taoamd_pred_copy(void *handle, int64_t *image_id, int64_t *category_id,
                     double *bbox, double *score, int64_t *track_id,
                     int64_t *video_id);
void taoamd_pred_free(void *handle);

/* ---- annotation file: {images, videos, tracks, annotations, categories, ...}
 * "not a dict" / "not a dict: list" when the document is not an object.
 * Arrays are fetched by name; the names are the fields of GTColumns
 * (tao_amodal_amd/columns.py):
 *   cat_id cat_freq(u8) cat_merged(pairs src,dst)
 *   vid_id vid_neg_off vid_neg vid_nel_off vid_nel            (CSR id lists)
 *   img_id img_vid img_frame(f64) img_neg_off img_neg img_nel_off img_nel
 *   trk_id trk_cat trk_vid trk_ignore(u8)
 *   ann_id ann_img ann_trk ann_cat ann_bbox(f64 x4) ann_area(f64) ann_vis(f64)
 *   ann_oof(u8) ann_ignore(u8)
 * "ignore" / "out_of_frame" follow Python truthiness of the JSON value.
 * taoamd_gt_array: *ptr stays valid until the handle is freed; *elem is 8
 * (int64), -8 (double) or 1 (uint8); returns nonzero for an unknown name. */
void *taoamd_gt_parse(const char *path, char *err, size_t errlen);
int taoamd_gt_array(void *handle, const char *name, const void **ptr,
                    int64_t *count, int *elem);
/* the named array copied into caller memory (count * |elem| bytes) by all cores */
int taoamd_gt_copy(void *handle, const char *name, void *dst);
void taoamd_gt_free(void *handle);

/* ---- the inverse: columns -> JSON files (synthetic sets written out for the
 * wall-clock runs of the CLI; csrc/jsonwrite.cpp).  Doubles are written as
 * the shortest text that parses back to the same value; track_id / video_id
 * may be NULL (keys omitted).  taoamd_gt_write takes the 28 arrays of
 * GTColumns.FIELDS in that order with their element counts.  Returns 0, 1 (bad
 * argument), 2 (cannot open), 3 (write error). */
int taoamd_pred_write(const char *path, int64_t n, const int64_t *image_id,
                      const int64_t *category_id, const double *bbox,
                      const double *score, const int64_t *track_id,
                      const int64_t *video_id);
int taoamd_gt_write(const char *path, const void *const *arrays,
                    const int64_t *counts, int32_t n_fields);

/* ---- host-side sort used while the cell tables are built
 * order[] = np.lexsort((arange(n), -score, key)) -- the (cell, descending
 * score, stable) order of lvis_amodal/eval.py:168-174 and the top-300
 * selection of lvis_amodal/results.py:73-84 -- as a parallel stable merge
 * sort; score may be NULL (stable argsort of key).  Returns 0. */
int taoamd_host_sort_key_score(int64_t n, const int64_t *key, const double *score,
                               int64_t *order);

/* ---- vector primitives of the ground-truth halves of the cell tables
 * (tao_amodal_amd/flatten.py: the id -> row resolutions of lvis_amodal/lvis.py:
 * 34-61 and tao_amodal/tao.py:108-160 over 3 M annotations), on all threads.
 * lookup: out[i] = index of values[i] in keys (ascending, unique) or -1.
 * take:   out[i] = src[idx[i]], elements of 1 / 4 / 8 / 32 bytes; returns 2
 *         when an index lies outside [0, n_src).
 * seq_mean: Python's left-to-right sum(...) / len(...) of each CSR segment
 *         (tao_amodal/tao.py:186-187).  All return 0 on success, 1 on bad
 *         arguments. */
int taoamd_host_lookup(int64_t n_keys, const int64_t *keys, int64_t n,
                       const int64_t *values, int64_t *out);
int taoamd_host_take(int32_t elem, int64_t n_src, const void *src, int64_t n,
                     const int64_t *idx, void *out);
int taoamd_host_seq_mean(int64_t n_seg, const int64_t *off, const double *vals,
                         double *out);

/* out[0 .. *n_out) = list(set(ids) & set(ids)) as CPython 3.7 - 3.12 iterates
 * it (ints 0 <= k < 2^61 - 1): the visiting order of the images of the
 * evaluated videos, tao_amodal/tao.py:224-230.  `out` holds n entries.  Returns
 * 0; 3 when a key is outside that range (the caller asks the interpreter). */
int taoamd_host_pyset_self_and(int64_t n, const int64_t *ids, int64_t *out,
                               int64_t *n_out);

/* *n_clash = how many track ids the predictions use in more than one video
 * (tools/eval_on_tao_amodal.py:44-58, len(track_ids_to_update)); 0 = nothing
 * to renumber, the usual case.  Returns 0; 3 when the ids span too wide a
 * range for a dense table (the caller's general statement takes over). */
int taoamd_host_track_clash(int64_t n, const int64_t *track_id, const int64_t *video_id,
                            int64_t *n_clash);

/* *n_bad = how many of the n boxes (x, y, w, h) have x < 0, y < 0, w <= 0 or
 * h <= 0: the count of the reference's "annotations had negative values in
 * coordinates" warning (tao_amodal/tao.py:143-158).  Returns 0. */
int taoamd_host_count_bad_boxes(int64_t n, const double *bbox, int64_t *n_bad);

/* OpenMP threads the host-side entry points of this library start: the
 * logical CPUs of the process capped by its affinity mask and by the control
 * group's CPU quota (csrc/host_threads.hpp; TAOAMD_HOST_THREADS overrides). */
int taoamd_host_threads(void);

/* Caps the OpenMP teams that entry points called FROM THIS host thread start
 * (n <= 0: no cap); returns the previous cap.  For callers that run several
 * entry points side by side under a CPU quota (tools/eval_on_tao_amodal.py: the
 * two readers beside the thread that imports torch in a fresh process). */
int taoamd_host_thread_cap(int n);

/* 1 if every one of values[0..n) occurs in keys[0..n_keys) (ascending), 0 if
 * one does not, -1 on a bad argument: the membership test behind "Results do
 * not correspond to current LVIS set." (reference lvis_amodal/results.py:62-65). */
int taoamd_host_all_in_sorted(int64_t n_keys, const int64_t *keys, int64_t n,
                              const int64_t *values);

/* ---- run-length masks (csrc/rle.cpp): the host side of iou_type="segm"
 * A batch collects masks in the order they are added and keeps them back to
 * back; taoamd_rle_copy hands out the CSR arrays the device kernel
 * taoamd_rle_iou (tao_amodal_hip.h) reads.  What the calls replace:
 *   add_polygons  LVIS.ann_to_rle on a polygon list: frPyObjects + merge
 *                 (lvis_amodal/lvis.py:171-186; pycocotools _mask.pyx frPoly /
 *                 merge over maskApi.c rleFrPoly:161-202, rleMerge:49-71).
 *                 part_off[n_parts+1] delimits the parts inside xy (in
 *                 doubles; a part of 2k or 2k+1 numbers has k vertices).
 *   add_counts    an uncompressed RLE {"size", "counts": [...]} (lvis.py:
 *                 187-189, frUncompressedRLE)
 *   add_string    a compressed RLE {"size", "counts": "..."} (lvis.py:190-192,
 *                 maskApi.c rleFrString:217-230)
 *   copy          also mask_utils.area / toBbox (lvis_amodal/results.py:56-59;
 *                 rleArea:72-75, rleToBbox:133-147): area[n], bbox[n][4]
 *                 (x, y, w, h); hw[n][2] = (height, width).  Any output
 *                 pointer may be NULL.
 *   string        the compressed text of one mask (rleToString:203-215) --
 *                 what ann["segmentation"]["counts"] holds after _to_mask;
 *                 returns its length, writes at most cap-1 characters + NUL.
 * add_* return the index of the new mask, or -1 for arguments no mask can be
 * made of (no part, a part without a vertex, a frame of 2^32 pixels or more).
 * A batch is not thread safe; different batches are independent. */
void *taoamd_rle_new(void);
void taoamd_rle_free(void *handle);
int64_t taoamd_rle_count(void *handle);
int64_t taoamd_rle_total(void *handle);           /* runs in all masks */
int64_t taoamd_rle_add_polygons(void *handle, int32_t n_parts,
                                const int64_t *part_off, const double *xy,
                                int64_t height, int64_t width);
/* n_masks polygon annotations at once (rasterised on all cores, appended in
 * order): mask m = parts [mask_part_off[m], mask_part_off[m+1]) of part_off,
 * frame hw[m] = (height, width); returns the index of the first new mask. */
int64_t taoamd_rle_add_polygon_batch(void *handle, int64_t n_masks,
                                     const int64_t *mask_part_off,
                                     const int64_t *part_off, const double *xy,
                                     const int32_t *hw);
int64_t taoamd_rle_add_counts(void *handle, const uint32_t *counts, int64_t m,
                              int64_t height, int64_t width);
int64_t taoamd_rle_add_string(void *handle, const char *text, int64_t height,
                              int64_t width);
int taoamd_rle_copy(void *handle, int64_t *off, uint32_t *counts, int32_t *hw,
                    uint32_t *area, double *bbox);
int64_t taoamd_rle_string(void *handle, int64_t index, char *buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
