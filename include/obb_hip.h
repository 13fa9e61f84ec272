/*
 * obb_hip.h -- C ABI of libobb_hip.so, the MI355X (gfx950) oriented-box hot path.
 *
 * Drop-in boundary for hukaixuan19970627/yolov5_obb: every entry point names the
 * reference interface it replaces (file:line, relative to the reference root).
 * Conventions shared by all functions:
 *   - plain C: raw pointers + sizes, no torch / ATen types;
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - buffers are caller-owned; inputs are never modified;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and is stream-ordered: no hidden host synchronisation, no
 *     allocation, no device->host copy.  Scratch memory comes from `ws`
 *     (size from the matching *_workspace_bytes query; 256-byte aligned);
 *   - return value: OBB_OK (0) or a negative OBB_ERR_* code; nothing throws.
 *     The Python host layer turns codes into RuntimeError, mirroring the
 *     reference's AT_ASSERTM / AT_ERROR -> RuntimeError behaviour
 *     (utils/nms_rotated/src/nms_rotated_ext.cpp:29-54).
 *   - floating point: IEEE fp32, no FMA contraction; results are defined by the
 *     CPU oracle in oracle/ (pinned to the reference's own sources).
 */
#ifndef OBB_HIP_H
#define OBB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OBB_OK 0
#define OBB_ERR_BAD_ARG (-1)
#define OBB_ERR_WORKSPACE (-2)   /* ws == NULL or ws_bytes too small */
#define OBB_ERR_LAUNCH (-3)      /* a kernel launch failed (see hipGetLastError) */
#define OBB_ERR_INTERNAL (-4)
#define OBB_ERR_NO_DEVICE (-5)

/* flags for obb_nms_rotated_* */
#define OBB_NMS_DROP_SMALL 1 /* ignore boxes with min(w,h) < 0.001 (utils/nms_rotated/nms_rotated_wrapper.py:32-39) */

/* Library / device identification: returns OBB_OK and fills the fields when a gfx950-class device is usable. */
const char* obb_version(void);
int obb_device_info(int* cu_count, int* wave_size, char* arch_name, int arch_name_len);
/* The persistent NMS kernel launches one workgroup per CU and its workgroups meet at spin barriers: they must all be
 * resident.  Where that cannot be taken for granted (CU masking, a partition that exposes fewer CUs than it reports, a
 * long kernel of another process holding CUs) a barrier times out and the call reports -1 kept boxes instead of hanging.
 * max_workgroups > 0 caps the grid of the following launches of the CALLING THREAD (thread-local state: the host layer
 * retries an aborted call once with 8 workgroups and restores the default in a finally block; concurrent callers on other
 * threads are not affected, and one call sizes its workspace and its grid from the same value); 0 restores the default
 * (the CU count). */
int obb_nms_set_max_grid(int max_workgroups);

/* Optional per-stage timing with HIP events recorded on the caller's stream (used by bench.py for the roofline
 * object).  Stage ids: 0 decode/filter kernel, 1 per-image sort, 2 candidate prep, 3 NMS steps, 4 output gather of
 * obb_non_max_suppression_obb; 5 sort, 6 prep, 7 NMS steps of obb_nms_*.  obb_profile_collect synchronises the
 * recorded events, returns summed milliseconds and launch counts per stage, and resets the recording.
 * on = 1: every stage (ten event records per fused call: ~30 us of a 0.2 ms step); on = 2: only the NMS kernels (stages 3 and 7,
 * the dominant kernel of a call: two records); 0: off. */
#define OBB_PROF_STAGES 8
int obb_profile_enable(int on);
int obb_profile_collect(double* ms_sum_host, int64_t* count_host, int n_stages);

/* Environment variables read by the library (once per process).  Every build:
 *   OBB_NMS_POLY_STRICT=1   quad NMS: only the proved skip rules (csrc/piou_device.h: the two cone rules); =2: no rule at all, every pair
 *                           is clipped like the reference does (tests/test_nms_gpu.py::test_nms_poly_strict_equals_skip_100k)
 *   OBB_NMS_PHASE_PROF=1    in-kernel phase timers of the single-list NMS (either path), printed to stderr (synchronises; development aid)
 * Read per call (tests switch paths inside one process; nothing but speed depends on them -- the kept list is the same on every path):
 *   OBB_NMS_MK=0 | 1 | 2    single-list rotated NMS of >= 16384 boxes: 0 the persistent kernel (csrc/nms_core.h), 1 the phase kernels
 *                           (csrc/nms_mk.h), 2 = default: the library's choice from what the calling thread's previous calls of the size
 *                           class reported (kept boxes, independent slabs, device time of either path)
 *   OBB_NMS_MK_XLDS=0 | 1   phase kernels: the cross probe on the chunk's table in global memory (0) or on a table of the kept rows in
 *                           every workgroup's LDS (1); default: 1 where the previous call kept <= 6144 boxes
 *   OBB_NMS_MK_NOFULL=0     phase kernels: the decide kernels run the IoU-interval stage in front of the exact clip again (default: the
 *                           pending pairs go straight to the clip)
 *   OBB_NMS_SELF_SORT=0|1|2 obb_non_max_suppression_obb* with the small-segment NMS kernel (expected_cand bits 32..60) and out_packed = 0:
 *                           0 = the per-image sort kernel in front of it, 1 = self-sorting segments (csrc/nmsobb_impl.h: SmallSelfSort --
 *                           no sort launch, two launches per call) where the sort kernel would hand out no helper workgroups,
 *                           2 = default: self-sorting segments wherever they are possible
 *   OBB_NMS_SMALL_HELPERS=n small-segment NMS kernel behind the sort kernel: helper workgroups that share large segments (0: none)
 * Read once per process, tests only: OBB_NMS_MK_STEPS (enqueued steps), OBB_NMS_MK_PEND (pending-list capacity), OBB_NMS_MK_CHUNK
 * (first chunk): they force the hand-overs to the persistent kernel that tests/test_nms_mk_gpu.py checks.
 * Development builds only (make DEV=1; ignored otherwise): the A/B switches OBB_NMS_NO_GRID, OBB_NMS_NO_SLABS, OBB_NO_CLASS_SEG,
 * OBB_NO_LDS_SORT, OBB_NMS_GROUP_AFTER_CUT, OBB_NMS_CHUNK*, OBB_NMS_GROW, OBB_NMS_SLAB_CAP, OBB_GRID_FINE, OBB_LOSS_NT. */

/* ------------------------------------------------------------------ NMS ------------------------------ */

/* Scratch bytes for n boxes in nseg segments.  kind: 0 = rotated boxes, 1 = quads, 2 = double-precision quads (merge NMS),
 * 3 = double-precision rotated boxes (obb_nms_rotated_f64), 4 = obb_merge_nms_poly_all_f64, 5 = obb_merge_nms_hbb_f64. */
size_t obb_nms_workspace_bytes(int64_t n, int64_t nseg, int kind);

/*
 * Rotated-box NMS.  Replaces nms_rotated_ext.nms_rotated on CUDA tensors
 * (utils/nms_rotated/src/nms_rotated_ext.cpp:25-39 -> nms_rotated_cuda,
 *  utils/nms_rotated/src/nms_rotated_cuda.cu:71-134: sort, N x N/64 mask kernel,
 *  device->host mask copy, host greedy scan).
 *   dets5    [n,5] fp32 contiguous  (cx, cy, w, h, angle in RADIANS)
 *   scores   [n]   fp32
 *   iou_thr  a box is dropped iff an earlier kept box has IoU > iou_thr (strict, cu:60)
 *   max_keep 0 = unlimited; otherwise stop after this many kept boxes (the caller's max_det,
 *            utils/general.py:854-855 truncates the same prefix)
 *   keep_out [n] int64: original indices of kept boxes in descending-score order
 *            (ties: ascending index; NaN scores first -- torch.sort's order)
 *   num_keep [1] int64 (device)
 * Lists of >= 16384 boxes without max_keep: two implementations of the same lazy chunked greedy NMS sit behind this entry -- a chain
 * of ordinary launches, one per phase (no co-residency needed), and one persistent kernel behind spin barriers (see
 * obb_nms_set_max_grid) -- and the library takes the one that was faster for the calling thread's previous calls of the size class
 * (four pinned words per size class that the device writes and the host reads without synchronising).  The persistent kernel is always
 * launched last: it completes whatever the enqueued phases left undone and returns at once otherwise.
 */
int obb_nms_rotated_f32(const float* dets5, const float* scores, int64_t n, float iou_thr, int flags, int64_t max_keep,
                        int64_t* keep_out, int64_t* num_keep, void* ws, size_t ws_bytes, void* stream);

/*
 * The same for float64 tensors: the reference dispatches double to a double-precision instantiation of the kernel
 * (AT_DISPATCH_FLOATING_TYPES, nms_rotated_cuda.cu:96; box_iou_rotated_utils.h:333-360 with T = double).  dets5 / scores
 * are doubles; every IoU is computed in double and compared with the FLOAT threshold of the kernel's signature
 * (nms_rotated_cuda.cu:14,60).  Workspace: obb_nms_workspace_bytes(n, 1, 3).
 */
int obb_nms_rotated_f64(const double* dets5, const double* scores, int64_t n, float iou_thr, int flags, int64_t max_keep,
                        int64_t* keep_out, int64_t* num_keep, void* ws, size_t ws_bytes, void* stream);

/*
 * Quadrilateral NMS.  Replaces nms_rotated_ext.nms_poly (nms_rotated_ext.cpp:42-55 -> poly_nms_cuda,
 * utils/nms_rotated/src/poly_nms_cuda.cu:197-261).  Rows are x1 y1 x2 y2 x3 y3 x4 y4 score (+ ignored extra
 * columns): row_stride >= 9 floats.  Kept set = the reference's greedy scan over devPolyIoU (poly_nms_cuda.cu:26-142).  Pairs whose
 * bounding boxes are disjoint AND whose areas outweigh the rounding noise of the reference's origin-based sum (a bound of
 * 1024 * 2^-24 * M^2 per box, M = its largest |coordinate|) are not clipped.  That bound is backed by search
 * (2.8 * 10^10 skip decisions, none wrong), not by proof, and is only applied inside the envelope the search covered: a quad with
 * every |coordinate| <= 70,000 and a bounding box of at most 600 x 600; any other quad is decided by the two PROVED rules (the two
 * quads' cones, as seen from the coordinate origin, apart by a margin in either order: all 16 terms exactly zero -- csrc/piou_device.h:
 * quad_cone_skip, quad_cone2_skip / quad_cone2_nofuzzy) or clipped like the reference does.
 * OBB_NMS_POLY_STRICT=1 keeps only the proved rules (4.5x the default's time at 30,000 quads), =2 clips every pair.
 */
int obb_nms_poly_f32(const float* polys, int64_t row_stride, int64_t n, float iou_thr, int64_t max_keep, int64_t* keep_out,
                     int64_t* num_keep, void* ws, size_t ws_bytes, void* stream);

/*
 * Tile -> full-image merge NMS in double precision.  Replaces py_cpu_nms_poly_fast
 * (DOTA_devkit/ResultMerge_multi_process.py:62-123; the same function in ResultMerge.py and
 * ResultEnsembleNMS_multi_process.py) for ALL images of a Task1_<class>.txt file in one call: the horizontal-box
 * gate (hbb_ovr > 0, :82-98) followed by DOTA_devkit/polyiou.cpp:108-128 (iou_poly) in IEEE double, a box
 * being dropped unless iou <= thresh (:115; a NaN IoU drops it).
 *   dets9    (n, 9) doubles  x1 y1 .. x4 y4 score  (only the 8 coordinates are read)
 *   order    (n) int32       row indices in processing order, segment after segment: the caller applies the reference's
 *                            `scores.argsort()[::-1]` (:79) per image, so score ties are ordered exactly like numpy's
 *   seg_off  (nseg + 1) int32  segment g = order[seg_off[g] .. seg_off[g+1])
 *   keep_out (n) int64       kept ROW indices of segment g at keep_out[seg_off[g] + k], k < num_keep[g], processing order
 *   num_keep (nseg) int64    (-1 in every entry: the device gave up on an internal barrier)
 * Workspace: obb_nms_workspace_bytes(n, nseg, 2).
 */
int obb_merge_nms_poly_f64(const double* dets9, int64_t n, const int32_t* order, const int32_t* seg_off, int64_t nseg,
                           double thresh, int64_t* keep_out, int64_t* num_keep, void* ws, size_t ws_bytes, void* stream);

/*
 * The other two merge variants of the same file, same conventions (order / seg_off / keep_out / num_keep as above):
 *   obb_merge_nms_poly_all_f64   py_cpu_nms_poly (ResultMerge_multi_process.py:24-60): no horizontal-box gate -- iou_poly of the
 *                                kept box and EVERY remaining candidate.  Workspace: obb_nms_workspace_bytes(n, nseg, 4).
 *   obb_merge_nms_hbb_f64        py_cpu_nms (:125-157; what mergebyrec hands to mergebase): horizontal boxes [x1 y1 x2 y2] in
 *                                columns 0..3 of rows of row_stride doubles, "+ 1" in areas and intersections, numpy double
 *                                arithmetic incl. its NaN rules (np.maximum hands a NaN through; 0 / 0 removes the box).
 *                                Workspace: obb_nms_workspace_bytes(n, nseg, 5).
 */
int obb_merge_nms_poly_all_f64(const double* dets9, int64_t n, const int32_t* order, const int32_t* seg_off, int64_t nseg,
                               double thresh, int64_t* keep_out, int64_t* num_keep, void* ws, size_t ws_bytes, void* stream);
int obb_merge_nms_hbb_f64(const double* dets, int64_t row_stride, int64_t n, const int32_t* order, const int32_t* seg_off,
                          int64_t nseg, double thresh, int64_t* keep_out, int64_t* num_keep, void* ws, size_t ws_bytes, void* stream);

/*
 * DOTA Task-1 evaluation: the detection x ground-truth part of voc_eval (DOTA_devkit/dota_evaluation_task1.py:168-223)
 * for all detections of a class file in one call.  Per detection, over the ground-truth quads of its own image: the
 * horizontal-box gate with the +1 convention (:181-204), iou_poly(GT, det) of DOTA_devkit/polyiou.cpp in IEEE double for
 * the quads that pass (:206-213), then np.max / np.argmax (:215-218).
 *   dets8   (nd, 8) doubles, in the caller's (confidence-sorted) order      det_img (nd) int32  image index of a detection
 *   gts8    (ng, 8) doubles, the quads of image i at gt_off[i] .. gt_off[i+1]   gt_off (n_img + 1) int32
 *   ovmax   (nd) doubles   -inf: no quad passed the gate (:172); NaN: some IoU was NaN
 *   jmax    (nd) int32     index within the image's list (first maximum; first NaN; -1 with -inf)
 * The sequential TP/FP marking (:225-233) is a first-occurrence pass over (image, jmax) and stays with the caller.
 */
int obb_eval_best_gt_f64(const double* dets8, const int32_t* det_img, int64_t nd, const double* gts8, const int32_t* gt_off,
                         int64_t n_img, double* ovmax, int32_t* jmax, void* stream);

/*
 * Host-side text I/O of the merge (no device work; csrc/textio.hip).  obb_task1_parse_tiles restates
 * DOTA_devkit/ResultMerge_multi_process.py:186-213 for a whole Task1_<class>.txt buffer: per line
 * `<orig>__<rate>__<x>___<y> score x1 y1 .. x4 y4` -> dets9[line] = [8 source-image coordinates (poly + x|y) / rate,
 * score] (strtod = Python's float()), the position of <orig> in the text, a group id per distinct <orig> in order of
 * first appearance and the first line of every group.  Returns the number of lines, or OBB_ERR_BAD_ARG for anything that
 * is not the plain layout (the Python layer then parses line by line like the reference).
 * obb_task1_format_rows writes `<orig> <round(score, 2)> <round(c, 1)> x 8\n` for the given lines (:218-233) and returns
 * the number of bytes (OBB_ERR_WORKSPACE: out_cap too small; 200 bytes per row + the name suffice for |values| < 1e15;
 * the text equals Python's str(round(..)) for |confidence| < 1e13 and |coordinates| < 1e14, the domain the Python layer checks).
 */
int64_t obb_task1_parse_tiles(const char* text_host, int64_t len, int64_t max_lines, double* dets9_host, int32_t* name_off_host,
                              int32_t* name_len_host, int32_t* group_host, int32_t* group_first_host, int64_t* n_groups_host);
/* The two readers of the Task-1 evaluation (dota_evaluation_task1.py): detections `image score x1 y1 .. x4 y4` (:152-160)
 * -> conf[line], bb8[line][8], position of the image id; ground truth `x1 .. y4 name [difficult]` (:21-53; lines with fewer
 * than 9 fields skipped, difficult = 0 when absent) -> bbox8, position of the class name, the flag.  Same contract as
 * obb_task1_parse_tiles: the number of records, or OBB_ERR_BAD_ARG for a buffer that is not the plain layout. */
int64_t obb_task1_parse_dets(const char* text_host, int64_t len, int64_t max_lines, double* conf_host, double* bb8_host,
                             int32_t* name_off_host, int32_t* name_len_host);
int64_t obb_task1_parse_gt(const char* text_host, int64_t len, int64_t max_lines, double* bbox8_host, int32_t* name_off_host,
                           int32_t* name_len_host, int32_t* difficult_host);
int64_t obb_task1_format_rows(const char* text_host, const int32_t* name_off_host, const int32_t* name_len_host,
                              const double* dets9_host, const int64_t* rows_host, int64_t n_rows, char* out_host, int64_t out_cap);

/* ------------------------------------------------------------------ fused NMS driver ----------------- */

/*
 * The whole of non_max_suppression_obb (utils/general.py:772-862) for a batch, in one stream-ordered call:
 * confidence filter, obj*cls, CSL decode (arg-max over the 180 angle bins -> theta = (idx-90)/180*3.141592),
 * multi-label expansion or best class, class filter, per-image top max_nms by confidence, class offset
 * (xy += cls*max_wh unless agnostic), obb_nms (incl. its min(w,h) < 0.001 filter), max_det truncation.
 *   pred        [bs][A][no] contiguous, no = 5 + nc + 180, rows [cx cy l s obj cls[nc] csl[180]] as produced by
 *               Detect's inference branch (models/yolo.py:67-81); dtype 0 = fp32, 1 = fp16 (val.py --half).
 *               The arithmetic follows the input dtype exactly like the reference (conf = obj*cls is rounded to
 *               fp16 for fp16 input, thresholds are compared in that dtype).
 *   classes_host  optional HOST array of allowed class ids (n_classes entries; NULL = all)     (:834-835)
 *   extra8      optional device rows [img, x, y, l, s, theta, conf, cls] appended as candidates: the apriori
 *               `labels` of autolabelling (:807-813), prepared by the host layer; n_extra rows
 *   cap_img     candidate slots reserved per image.  If an image produces more, status[0] receives that count
 *               (> cap_img) and the caller must retry with a larger cap_img (A*nc can never overflow).
 *   expected_cand  what the caller expects, from status[1] of its previous call of this shape (0 = unknown: always valid):
 *               bits 0..31  candidates per image.  Selects the sort: 1 .. OBB_NMS_SORT_LDS_HINT -> the images are sorted in LDS
 *               by a kernel that also builds the NMS records (six launches fewer); larger -> the multi-workgroup sorts of
 *               csrc/segsort.h / csrc/psrs_sort.h.  The in-LDS sort takes at most OBB_NMS_SORT_LDS_MAX candidates of an image:
 *               with a hint in 1 .. OBB_NMS_SORT_LDS_HINT and status[1] (low word) > OBB_NMS_SORT_LDS_MAX after the call, such
 *               images were left EMPTY -- call again with that count as the hint (the Python layer does).
 *               bits 32..60 the largest NMS segment (boxes of one class of one image).  1 .. OBB_NMS_SMALL_SEG together with the
 *               in-LDS sort and iou_thres >= 0 selects the one-workgroup-per-segment NMS kernel (csrc/nms_small.h); a call that
 *               meets a larger segment there sets status[0] = -1: NOTHING of its output is valid, call again with the
 *               segment size status[1] reports (the Python layer does).  0 or larger: the persistent kernel.
 *               With out_packed = 0 that kernel's (image, class) workgroups also ORDER their class themselves (round 6: no sort
 *               launch; OBB_NMS_SELF_SORT above) and the reported size is the largest class of any image of up to
 *               OBB_NMS_SORT_LDS_MAX candidates; behind the sort kernel an image above OBB_NMS_SORT_LDS_MAX / 2 candidates is
 *               ordered as one list and reported with its whole size.
 *               bit 62      the previous call met boxes with a short side in [0.001, 1) px (status[1] bit 62).  The reference's fp32
 *               clip is ill conditioned for such a box against a partner tens of thousands of pixels away, i.e. across its
 *               cls * max_wh offsets (utils/general.py:849-851), so an image holding one may only keep its per-class NMS segments if
 *               no such pair has IoU > iou_thres.  With the bit set the call checks exactly that (k_tiny_cross: the reference's own
 *               arithmetic on every ill-conditioned cross-class pair; images with at most 64 such boxes and 262,144 box x candidate
 *               combinations: microseconds for the stray sub-pixel box of a trained detector) and keeps the class segments of every
 *               image that passes; without it -- and for an image that fails, is above those bounds, or holds a box whose circle
 *               leaves a 0.95 max_wh window -- the image runs as the reference's single list.  Either way the rows are the reference's.
 *               Apart from the two retry cases the result does not depend on the hint.
 *   out         [bs][max_det][7] fp32 rows [x y l s theta conf cls];  out_count [bs] int64 (-1: device-side abort);
 *               out_packed != 0: the rows of image b start right behind those of image b-1 (row sum(out_count[0..b-1]))
 *               instead of at row b*max_det -- the same buffer size is required, one split instead of bs slices on the host.
 *               out_packed = 0 is the faster form where the small-segment NMS kernel runs (expected_cand, bits 32..60): the rows of an
 *               image do not wait for the counts of the images in front of it, so that kernel writes them itself (csrc/nmsobb_impl.h:
 *               SmallGather) and the call is one launch shorter; obb_val_tail_batch_rows_f32 consumes that layout in place.
 *               status [2] int64: [0] overflow count (see cap_img), or -1 (see expected_cand); [1] largest candidate count of any
 *               image in bits 0..31, largest NMS segment in bits 32..60 (0 where the sort path does not know it), bit 62: an image
 *               held boxes with a sub-pixel short side -- hand these back as expected_cand of the next call of the shape; bit 61
 *               (informational, masked out of the hint): such an image kept its class segments in this call
 *               out_count and status are written with plain 8-byte stores by the last kernel of the call: they may live in
 *               device memory or in pinned host memory (hipHostMalloc) that the caller polls instead of copying back.
 *               What a polled word publishes is the COUNT, nothing else: the kernel that writes out_count[b] / status may still be
 *               storing rows of `out` (of image b and of the images in flight) and reading `ws` at that moment.  `out` is complete and
 *               `ws` is reusable in STREAM ORDER only: a consumer on the call's stream needs nothing more; a consumer on another stream
 *               or on the host records / waits for an event behind the call (or synchronises the stream) before it touches `out` or
 *               hands `ws` to anybody else.  (The bindings of this repository return views of `out` to same-stream consumers and keep
 *               `ws` in a per-stream cache.)
 * Score ties are ordered by ascending (anchor*nc + class): deterministic, where the reference inherits the order
 * of torch's unstable sort.
 */
#define OBB_NMS_SMALL_SEG 384
#define OBB_NMS_SORT_LDS_HINT 6144
#define OBB_NMS_SORT_LDS_MAX 8192
size_t obb_nms_obb_workspace_bytes(int64_t bs, int64_t cap_img, int64_t nc, int agnostic);
int obb_non_max_suppression_obb(const void* pred, int dtype, int64_t bs, int64_t A, int64_t no, float conf_thres,
                                float iou_thres, const int32_t* classes_host, int n_classes, int agnostic, int multi_label,
                                int64_t max_det, int64_t max_nms, float max_wh, const float* extra8, int64_t n_extra,
                                int64_t cap_img, int64_t expected_cand, float* out, int out_packed, int64_t* out_count,
                                int64_t* status, void* ws, size_t ws_bytes, void* stream);
/* The same call for a `pred` whose producer also stored the objectness column densely: objcol [bs][A] in pred's dtype,
 * objcol[b][i] == pred[b][i][4] bit for bit (obb_detect_decode_col writes it next to z).  The confidence filter
 * (utils/general.py:785 xc = prediction[..., 4] > conf_thres) then reads bs*A elements instead of one 128-byte line of
 * every 400-800-byte row; rows that pass are read from `pred` as before, so the result is the one of the plain call.
 * objcol == NULL is the plain call. */
int obb_non_max_suppression_obb_col(const void* pred, const void* objcol, int dtype, int64_t bs, int64_t A, int64_t no,
                                    float conf_thres, float iou_thres, const int32_t* classes_host, int n_classes, int agnostic,
                                    int multi_label, int64_t max_det, int64_t max_nms, float max_wh, const float* extra8,
                                    int64_t n_extra, int64_t cap_img, int64_t expected_cand, float* out, int out_packed,
                                    int64_t* out_count, int64_t* status, void* ws, size_t ws_bytes, void* stream);

/* The same call with its candidate counters in a buffer of the CALLER's that outlives the call: `state` = obb_nms_obb_state_bytes(bs)
 * bytes of device memory (256-byte aligned), zeroed ONCE by the caller (hipMemset) and then handed to every call with this bs
 * that is ordered on one stream.  Each call finds it zeroed and leaves it zeroed (its last kernel does that), so the call has no
 * reset launch of its own; the filter kernel zeroes the state of the later launches on its way.  At the reference's default
 * thresholds with out_packed = 0 the whole of non_max_suppression_obb is then TWO launches (filter + CSL decode; NMS whose
 * (image, class) workgroups order their own class and write the output rows -- round 6, OBB_NMS_SELF_SORT above; three with the sort
 * kernel between them).  A call that returns an error may leave the buffer dirty: zero it again.  Two calls that are not ordered on one
 * stream need a buffer each.  Everything else as obb_non_max_suppression_obb_col (objcol may be NULL). */
size_t obb_nms_obb_state_bytes(int64_t bs);
int obb_non_max_suppression_obb_st(const void* pred, const void* objcol, int dtype, int64_t bs, int64_t A, int64_t no,
                                   float conf_thres, float iou_thres, const int32_t* classes_host, int n_classes, int agnostic,
                                   int multi_label, int64_t max_det, int64_t max_nms, float max_wh, const float* extra8,
                                   int64_t n_extra, int64_t cap_img, int64_t expected_cand, float* out, int out_packed,
                                   int64_t* out_count, int64_t* status, void* ws, size_t ws_bytes, void* state, size_t state_bytes,
                                   void* stream);

/* ------------------------------------------------------------------ training loss -------------------- */

/*
 * ComputeLoss of the OBB head (utils/loss.py:90-275): build_targets + box (horizontal CIoU) / objectness / class /
 * CSL-angle losses, forward and backward, for raw head outputs p[i] of shape (bs, na, ny_i, nx_i, no) and the
 * dataloader's target tensor (nt, 7+180) rows [img, cls, cx, cy, l, s, theta, csl x 180] in pixels
 * (utils/datasets.py:637-672) -- or (nt, 7) rows without the labels, see obb_loss_config.csl_radius.  Everything ComputeLoss.__init__ reads from the model (utils/loss.py:93-120) travels
 * in obb_loss_config (host memory, plain C).
 */
#define OBB_LOSS_MAX_LEVELS 8
#define OBB_LOSS_MAX_ANCHORS 8
typedef struct obb_loss_config {
  int32_t nl, na, nc, no, bs;          /* Detect.nl / na / nc / no (= 5 + nc + 180), batch size            */
  int32_t ny[OBB_LOSS_MAX_LEVELS], nx[OBB_LOSS_MAX_LEVELS];     /* p[i].shape[2], p[i].shape[3]             */
  float anchors[OBB_LOSS_MAX_LEVELS][OBB_LOSS_MAX_ANCHORS][2];   /* Detect.anchors (grid units), (l, s)      */
  float stride[OBB_LOSS_MAX_LEVELS];                             /* Detect.stride                            */
  float balance[OBB_LOSS_MAX_LEVELS];                            /* utils/loss.py:114                        */
  float anchor_t;                      /* hyp['anchor_t']                                                   */
  float cp, cn;                        /* smooth_BCE(label_smoothing), utils/loss.py:104                    */
  float cls_pw, theta_pw, obj_pw;      /* BCE pos_weights, utils/loss.py:98-100                             */
  float gain_box, gain_obj, gain_cls, gain_theta;   /* hyp['box'|'obj'|'cls'|'theta'], utils/loss.py:185-188 */
  float gr;                            /* iou ratio, utils/loss.py:116                                      */
  int32_t sort_obj_iou;                /* utils/loss.py:93,156-158                                          */
  float csl_radius;                    /* hyp['csl_radius'] (data/hyps/obb/ yaml files; <= 0: 2.0).  Used only when the target
                                          rows carry no CSL labels (tcols == 7): the 180-bin label is then regenerated on
                                          the device from theta exactly as gaussian_label_cpu rolls its window
                                          (utils/rboxs_utils.py:9-26, utils/datasets.py:639-642) -- SURVEY 8(f) row 3   */
  float fl_gamma;                      /* hyp['fl_gamma']: > 0 wraps the class, angle and objectness BCE in FocalLoss(gamma,
                                          alpha = 0.25) like utils/loss.py:107-110 (35-62); 0: plain BCE                */
} obb_loss_config;

size_t obb_loss_workspace_bytes(const obb_loss_config* cfg, int64_t nt);

/* ComputeLoss.build_targets (utils/loss.py:194-275) into the workspace; counts_out[0..nl) (device, int32) receives
 * the number of matched rows per level and counts_out[OBB_LOSS_MAX_LEVELS] a non-zero flag when a target row names
 * an image or class outside the batch (the reference raises IndexError there).  No host synchronisation. */
int obb_loss_build_targets(const obb_loss_config* cfg, const float* targets, int64_t nt, int64_t tcols, int32_t* counts_out,
                           void* ws, size_t ws_bytes, void* stream);
/* Rows of one level in the reference's order (offset-major, anchor-major, target order), after the caller has read
 * the count n: indices4 [n][4] int64 (b, a, gj, gi); tbox4 [n][4]; anch2 [n][2]; tcls [n] int64; csl180 [n][180]. */
int obb_loss_export_targets(const obb_loss_config* cfg, int64_t nt, int level, int64_t n, int64_t* indices4, float* tbox4,
                            float* anch2, int64_t* tcls, float* csl180, void* ws, size_t ws_bytes, void* stream);

/* ComputeLoss.__call__ forward (utils/loss.py:122-192).  p_levels_host: HOST array of nl device pointers;
 * dtype 0 = fp32, 1 = fp16 (arithmetic is fp32 either way).  loss_out (device, 5 + nl floats):
 * [0] (lbox+lobj+lcls+ltheta)*bs, [1..4] lbox, lobj, lcls, ltheta (gains applied), [5+i] the un-balanced objectness
 * BCE of level i (what autobalance reads, :180-181).  [0] is NaN when a target row is out of range.
 * The workspace keeps the matched rows and (round 5) a dense copy of every anchor row's objectness logit for obb_loss_backward
 * (which then reads 4 contiguous bytes per row instead of one 128-byte line of each 800-byte row): pass the same buffer, untouched. */
int obb_loss_forward(const obb_loss_config* cfg, const void* const* p_levels_host, int dtype, const float* targets, int64_t nt,
                     int64_t tcols, float* loss_out, void* ws, size_t ws_bytes, void* stream);
/* Gradient of loss_out[0] with respect to every p[i], times *grad_scale (device scalar: the incoming dL/dloss, e.g.
 * the GradScaler factor).  grad_levels_host: HOST array of nl device pointers, same shapes / dtype as p; every
 * element is written exactly once (no prior zero-fill needed). */
int obb_loss_backward(const obb_loss_config* cfg, const void* const* p_levels_host, int dtype, const float* targets, int64_t nt,
                      int64_t tcols, const float* grad_scale, void* const* grad_levels_host, void* ws, size_t ws_bytes,
                      void* stream);

/* ------------------------------------------------------------------ Detect head / CSL / box utils ----- */

/*
 * Detect.forward, inference branch, for ONE level (models/yolo.py:61-79): conv_out is the 1x1-conv output
 * (bs, na*no, ny, nx), contiguous, dtype 0 = fp32 / 1 = fp16.  Writes (either pointer may be NULL)
 *   x_perm_out  (bs, na, ny, nx, no)  the raw head, what `x[i].view(bs,na,no,ny,nx).permute(0,1,3,4,2).contiguous()` returns (:65)
 *   z_out       rows [a_offset, a_offset + na*ny*nx) of every image of the concatenated prediction tensor
 *               (bs, a_total, no): sigmoid, xy = (y*2-0.5+grid)*stride, wh = (y*2)^2*anchor_grid (:71-79),
 *               arithmetic and rounding steps in the tensor dtype exactly as the reference's in-place branch.
 *   anchors_px_host  HOST array [na][2] = Detect.anchors[i] * stride[i] (anchor_grid, :90-91)
 */
int obb_detect_decode(const void* conv_out, int dtype, int64_t bs, int64_t na, int64_t no, int64_t ny, int64_t nx,
                      const float* anchors_px_host, float stride, void* x_perm_out, void* z_out, int64_t a_total,
                      int64_t a_offset, void* stream);
/* obb_detect_decode that also writes objcol_out [bs][a_total] (same dtype) = z[..., 4] of the level's rows: the column the
 * confidence filter of obb_non_max_suppression_obb_col reads.  Any of x_perm_out / z_out / objcol_out may be NULL. */
int obb_detect_decode_col(const void* conv_out, int dtype, int64_t bs, int64_t na, int64_t no, int64_t ny, int64_t nx,
                          const float* anchors_px_host, float stride, void* x_perm_out, void* z_out, int64_t a_total,
                          int64_t a_offset, void* objcol_out, void* stream);

/* The whole inference branch of Detect.forward (models/yolo.py:61-79: the loop over the nl levels and the torch.cat) in ONE
 * launch: conv_out[l] is level l's conv output (bs, na*no, ny[l], nx[l]); x_perm_out[l] its permuted raw head (the array or
 * any entry may be NULL); the levels' rows follow each other in z_out / objcol_out (bs, a_total, ...) in level order, like
 * torch.cat(z, 1).  anchors_px_host HOST [nl][na][2], strides_host HOST [nl].  nl <= 4.  Same bytes as nl calls of
 * obb_detect_decode_col with a_offset = the rows of the levels before. */
int obb_detect_decode_levels(int nl, const void* const* conv_out, int dtype, int64_t bs, int64_t na, int64_t no, const int64_t* ny,
                             const int64_t* nx, const float* anchors_px_host, const float* strides_host, void* const* x_perm_out,
                             void* z_out, int64_t a_total, void* objcol_out, void* stream);

/* gaussian_label_cpu (utils/rboxs_utils.py:9-26) for n angles at once: out [n][num_class] fp32, evaluated in double. */
int obb_csl_encode_f32(const float* labels, int64_t n, int num_class, double u, double sig, float* out, void* stream);

/* rbox2poly (utils/rboxs_utils.py:106-145) and poly2hbb of the result (:147-181): rows [cx cy l s theta ...] with
 * row_stride >= 5 floats -> poly8 [n][8] and/or hbb4 [n][4] = [xc yc w h] (either may be NULL). */
int obb_rbox2poly_f32(const float* rboxes, int64_t n, int64_t row_stride, float* poly8, float* hbb4, void* stream);

/* ------------------------------------------------------------------ post-NMS tail of val.py --------- */

/* val.py:226-236 for the detections of one image (n,7) [x y l s theta conf cls] in one kernel: pred_poly (n,10),
 * pred_hbb (n,6) in model-input space, pred_polyn (n,10) / pred_hbbn (n,6) in native image space
 * (scale_polys, utils/general.py:636-650: (x - pad_x) / gain, (y - pad_y) / gain).  Any output may be NULL. */
int obb_val_postprocess_f32(const float* det7, int64_t n, float pad_x, float pad_y, float gain, float* poly10, float* hbb6,
                            float* polyn10, float* hbbn6, void* stream);

/* The same tail for ALL images of a batch in two launches (val.py:209-250 is a per-image loop): det7 = the packed (N,7)
 * detections of the batch, image b's rows at [det_off_host[b], det_off_host[b+1]) (bs + 1 host integers, det_off_host[0] = 0,
 * bs <= 64); targets = the batch's labels on the device, (nt, tcols >= 7) rows [img cls cx cy l s theta ...] in pixels of the
 * letterboxed frame (the collate format); img5_host = bs x {pad_x, pad_y, gain, native width, native height} (shapes[si] of
 * val.py).  Label boxes go through rbox2poly -> poly2hbb -> xywh2xyxy -> scale_coords in the reference's order (val.py:238-241).
 * Outputs (device): the four packed arrays of obb_val_postprocess_f32 (any may be NULL) and stats (N, niou + 2) floats: the
 * correct row of process_batch as 0 / 1, then conf, then cls -- val.py:250's tuple in one array, one copy per batch. */
size_t obb_val_tail_batch_workspace_bytes(int64_t n_det, int64_t nt);
int obb_val_tail_batch_f32(const float* det7, const int64_t* det_off_host, int64_t bs, const float* targets, int64_t nt, int64_t tcols,
                           const float* img5_host, const float* iouv, int niou, float* poly10, float* hbb6, float* polyn10,
                           float* hbbn6, float* stats, void* ws, size_t ws_bytes, void* stream);
/* The same, for a caller that does not want to copy `stats` back and wait for the stream: `stats` and `done` point into PINNED
 * HOST memory (hipHostMalloc: the device writes through the same pointers); `done` receives n (the number of detections) once
 * every row of `stats` is visible to the host -- the caller sets it to a negative value before the call and polls it. */
int obb_val_tail_batch_polled_f32(const float* det7, const int64_t* det_off_host, int64_t bs, const float* targets, int64_t nt,
                                  int64_t tcols, const float* img5_host, const float* iouv, int niou, float* poly10, float* hbb6,
                                  float* polyn10, float* hbbn6, float* stats, void* ws, size_t ws_bytes, void* stream, int64_t* done);
/* The same for detections that are NOT packed: image b's rows are det7[det_row_host[b] .. det_row_host[b] + det_off_host[b+1] -
 * det_off_host[b]) (bs host integers, rows of 7 floats from det7) -- the layout obb_non_max_suppression_obb writes with
 * out_packed = 0 (det_row_host[b] = b * max_det), consumed where it lies.  Every OUTPUT stays packed by det_off_host.  done may be
 * NULL (then the caller waits for the stream), else as above. */
int obb_val_tail_batch_rows_f32(const float* det7, const int64_t* det_row_host, const int64_t* det_off_host, int64_t bs, const float* targets,
                                int64_t nt, int64_t tcols, const float* img5_host, const float* iouv, int niou, float* poly10, float* hbb6,
                                float* polyn10, float* hbbn6, float* stats, void* ws, size_t ws_bytes, void* stream, int64_t* done);

/* process_batch (val.py:69-90): detections (n,6) [x1 y1 x2 y2 conf cls], labels (m,5) [cls x1 y1 x2 y2], iouv (niou) on the
 * device -> correct (n, niou) bytes (0/1).  No device->host round trip (the reference sorts the matches with numpy). */
size_t obb_process_batch_workspace_bytes(int64_t n, int64_t m);
int obb_process_batch_f32(const float* det6, int64_t n, const float* lab5, int64_t m, const float* iouv, int niou, uint8_t* correct,
                          void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ pairwise IoU --------------------- */

/* out[i] = IoU(a5[i], b5[i]); the device function behind the NMS
 * (single_box_iou_rotated<float>, utils/nms_rotated/src/box_iou_rotated_utils.h:333-360). */
int obb_rotated_iou_pairs_f32(const float* a5, const float* b5, int64_t n, float* out, void* stream);
/* out[i*k + j] = IoU(a5[i], b5[j]) */
int obb_rotated_iou_matrix_f32(const float* a5, int64_t n, const float* b5, int64_t k, float* out, void* stream);
/* out[i*k + j] = quad IoU of rows (first 8 floats used) -- devPolyIoU, utils/nms_rotated/src/poly_nms_cuda.cu:122-142.
 * Entries the two proved cone rules of csrc/piou_device.h vouch for are written as the exact +0 the clip would return. */
int obb_quad_iou_matrix_f32(const float* a, int64_t a_stride, int64_t n, const float* b, int64_t b_stride, int64_t k,
                            float* out, void* stream);
/* Dense IoU matrix of rboxes through RotBox2Poly + devPolyIoU: the overlaps_kernel of
 * DOTA_devkit/poly_nms_gpu/poly_overlaps_kernel.cu:280-353 on device pointers. */
int obb_rbox_overlaps_f32(const float* boxes5, int64_t n, const float* query5, int64_t k, float* out, void* stream);

/* ------------------------------------------------------------------ devkit host-pointer API ---------- */

/* Same symbols, argument order and host-pointer convention as the reference's devkit
 * (DOTA_devkit/poly_nms_gpu/poly_nms.hpp:9-10, poly_overlaps.hpp:1); synchronous.  The reference declares the two
 * functions with C++ linkage: the library also exports them under the mangled names a build of the reference's Cython
 * sources binds (_Z9_poly_nmsPiS_PKfiifi, _Z9_overlapsPfPKfS1_iii).  tests/test_devkit_binding.py calls both spellings through
 * ctypes, and oracle/ref_shim_devkit.cpp -- a translation unit that includes the reference's own poly_nms.hpp / poly_overlaps.hpp
 * -- is linked against this library by oracle/Makefile and driven like poly_nms.pyx:9-24 / poly_overlaps.pyx:7-12. */
void _poly_nms(int* keep_out_host, int* num_out_host, const float* polys_host, int polys_num, int polys_dim,
               float nms_overlap_thresh, int device_id);
void _overlaps(float* overlaps_host, const float* boxes_host, const float* query_boxes_host, int n, int k, int device_id);

#ifdef __cplusplus
}
#endif
#endif /* OBB_HIP_H */
