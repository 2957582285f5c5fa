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
void taoamd_gt_free(void *handle);

/* ---- host-side sort used while the cell tables are built
 * order[] = np.lexsort((arange(n), -score, key)) -- the (cell, descending
 * score, stable) order of lvis_amodal/eval.py:168-174 and the top-300
 * selection of lvis_amodal/results.py:73-84 -- as a parallel stable merge
 * sort; score may be NULL (stable argsort of key).  Returns 0. */
int taoamd_host_sort_key_score(int64_t n, const int64_t *key, const double *score,
                               int64_t *order);

#ifdef __cplusplus
}
#endif
#endif
