/*
 * frcnn_hip.h -- C ABI of libfrcnn_hip.so: the MI355X (gfx950) kernels behind the Faster R-CNN
 * region-proposal / RoI hot path of DingLuBJUT/ObjectDetection (code/FasterRCNN).
 *
 * The reference is pure Python and has no FFI of its own: each entry point below
 * replaces the torch / torchvision call sequence at the cited reference lines (paths relative to
 * code/FasterRCNN/) and is what the Python drop-in package `objectdetection_amd` binds with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *   - the caller owns every buffer, workspace included (query the *_workspace_bytes functions);
 *     no entry point allocates, frees or synchronises -- all work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream);
 *   - fp32 everywhere, boxes are [x1, y1, x2, y2], index outputs are int32 unless stated;
 *   - return value: 0 on success, a negative FRCNN_E* code otherwise (never throws);
 *     frcnn_last_error_string() describes the last failure on the calling thread;
 *   - IoU is evaluated in fp32 exactly as utils.py:119 associates it,
 *     inter / ((area_a + area_b) - inter), with IEEE division and no FMA contraction.
 */
#ifndef FRCNN_HIP_H
#define FRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRCNN_OK 0
#define FRCNN_EINVAL (-1)    /* bad argument (shape, null pointer, size limit) */
#define FRCNN_EWORKSPACE (-2) /* workspace too small */
#define FRCNN_ELAUNCH (-3)   /* HIP reported a launch error */

int frcnn_abi_version(void);
const char *frcnn_last_error_string(void);

/* ---- K1 anchors ------------------------------------------------------------------------------
 * RPN/anchor_generator.py:43-64 (get_base_anchor).  HOST arithmetic on 4*na floats:
 * h = size, w = size*ratio (fp32), out = [-w,-h,w,h]/2.  All pointers are host pointers. */
int frcnn_base_anchors_f32(const float *size_host, const float *ratio_host, int na, float *out_host);

/* RPN/anchor_generator.py:66-95 (generate_anchor), one pyramid level:
 * out[(y*fw+x), a, :] = [x*sx, y*sy, x*sx, y*sy] + base[a, :]   (two fp32 roundings).
 * base: (na,4) device.  out: (fh*fw, na, 4). */
int frcnn_anchor_grid_f32(const float *base, int na, int fh, int fw, float stride_y, float stride_x,
                          float *out, void *stream);

/* ---- K2 flatten + top-k ----------------------------------------------------------------------
 * RPN/rpn.py:56-57: cls (B,na,h,w) -> scores[b, level_offset + (y*w+x)*na + a];
 * reg (B,4na,h,w) -> deltas[b, level_offset + (y*w+x)*na + a, 0:4].  One call per level.
 * scores: (B, A_total), deltas: (B, A_total, 4). */
int frcnn_rpn_flatten_f32(const float *cls, const float *reg, int B, int na, int h, int w,
                          int level_offset, int A_total, float *scores, float *deltas, void *stream);

/* RPN/rpn.py:65-66: per-row torch.topk(score, k), sorted.  Order: descending score, ties by
 * ascending index; NaN ranks above +inf; -0.0 below +0.0 (torch leaves tie order undefined).
 * scores: (B, n).  val_out: (B, k) fp32 (may be NULL).  idx_out: (B, k) int32.  k <= n. */
size_t frcnn_topk_workspace_bytes(int B, int n, int k);
int frcnn_topk_desc_f32(const float *scores, int B, int n, int k, float *val_out, int32_t *idx_out,
                        void *workspace, size_t workspace_bytes, void *stream);

/* The same with a cap on the workgroups an image is split over (max_slices; 0 = the library's choice:
 * images of more than 8,192 scores are cut into slices that are sorted by one workgroup each and then
 * merged, about one workgroup per CU; 1 = one workgroup per image).  A caller that runs top-k beside a
 * kernel which fills the chip's LDS (two batches in flight: objectdetection_amd/pipeline.py) keeps the
 * split small, because every slice workgroup needs a CU of its own.  No counterpart in the reference
 * (torch.topk, RPN/rpn.py:65-66); results do not depend on max_slices. */
int frcnn_topk_desc_ex_f32(const float *scores, int B, int n, int k, float *val_out,
                           int32_t *idx_out, void *workspace, size_t workspace_bytes,
                           int max_slices, void *stream);

/* ---- K3 decode / clip / small-box filter -----------------------------------------------------
 * utils.py:173-208 (box_regression, incl. quirk Q1: centre = (w/2, h/2)).
 * box: (n,4); regs: (n,4K) as [dx,dy,dw,dh]xK; out: (n,4K) as [x1,y1,x2,y2]xK. */
int frcnn_box_regression_f32(const float *box, const float *regs, int n, int K, float *out,
                             void *stream);

/* Backward of the decode above: what torch autograd produces through utils.py:173-208 (the
 * reference keeps its proposals on the autograd graph, RPN/rpn.py:75,85).  grad_out, grad_regs:
 * (n,4K); grad_box: (n,4) or NULL.  The clamp passes the gradient where dw <= log(1000/16). */
int frcnn_box_regression_backward_f32(const float *box, const float *regs, const float *grad_out,
                                      int n, int K, float *grad_regs, float *grad_box, void *stream);

/* utils.py:242-259 (box_clip): x to [0,image_w], y to [0,image_h] (inclusive). */
int frcnn_box_clip_f32(const float *box, int n, float image_h, float image_w, float *out,
                       void *stream);

/* utils.py:231-239 (remove_small_box, size part): keeps rows with w > t AND h > t in ascending
 * order.  box_out: (n,4) (first *count rows valid); index_out: (n) int64; count_out: (1) int32. */
int frcnn_remove_small_box_f32(const float *box, int n, float size_thresh, float *box_out,
                               int64_t *index_out, int32_t *count_out, void *stream);

/* utils.py:227-229 / ROIHead/roi_head.py:197,209: index = where(value > thresh)[0], ascending.
 * index_out: (n) int64; count_out: (1) int32. */
int frcnn_select_gt_f32(const float *value, int n, float thresh, int64_t *index_out,
                        int32_t *count_out, void *stream);

/* Fused RPN/rpn.py:67-69,75,81-83 for a whole batch: gather anchor+delta by top-k index, decode,
 * clip to (image_h,image_w), drop boxes with w<=t or h<=t, order preserved.
 * anchors: (A,4) shared by the batch when anchor_batch_stride == 0, else (B,A,4) with that
 * element stride; deltas: (B,A,4); top_idx/top_score: (B,k).
 * boxes_out: (B,k,4); scores_out: (B,k); src_out: (B,k) position in the top-k list of each kept
 * row (= remove_small_box's index); count_out: (B) kept rows per image.  Rows at and beyond an image's
 * count are unspecified (src_out may hold -1 there). */
int frcnn_decode_clip_filter_f32(const float *anchors, int64_t anchor_batch_stride,
                                 const float *deltas, const int32_t *top_idx,
                                 const float *top_score, int B, int A, int k, float image_h,
                                 float image_w, float min_size, float *boxes_out,
                                 float *scores_out, int32_t *src_out, int32_t *count_out,
                                 void *stream);

/* RPN/rpn.py:56-57 + 65-66 and 67-69,75,81-83 WITHOUT the flattened copies: the same two stages reading the
 * head's NCHW outputs where they lie.  cls_host / reg_host: HOST arrays of n_levels (<= 6) device pointers,
 * (B,na,h,w) scores and (B,4na,h,w) deltas; hw_host: HOST array (n_levels,2) of (h,w); flat element index
 * = level offset + cell*na + a, as frcnn_rpn_flatten_f32 lays it out, A = na * sum(h*w).
 * frcnn_topk_desc_nchw_f32 covers the sizes the split top-k covers (frcnn_topk_desc_nchw_supported != 0:
 * A > 8192 and max_slices != 1); other sizes flatten first.  Results are bit-identical to the flattened
 * path.  workspace: frcnn_topk_workspace_bytes(B, A, k). */
int frcnn_topk_desc_nchw_supported(int B, int A, int k, int max_slices);
int frcnn_topk_desc_nchw_f32(const float *const *cls_host, const int32_t *hw_host, int n_levels, int na,
                             int B, int k, float *val_out, int32_t *idx_out, void *workspace,
                             size_t workspace_bytes, int max_slices, void *stream);
int frcnn_decode_clip_filter_nchw_f32(const float *anchors, int64_t anchor_batch_stride,
                                      const float *const *reg_host, const int32_t *hw_host,
                                      int n_levels, int na, const int32_t *top_idx,
                                      const float *top_score, int B, int k, float image_h,
                                      float image_w, float min_size, float *boxes_out,
                                      float *scores_out, int32_t *src_out, int32_t *count_out,
                                      void *stream);

/* ---- the proposal half in one call -------------------------------------------------------------
 * RPN/rpn.py:54-85 for the whole batch: top-k of the head's NCHW scores (:65-67), decode + clip + small-box filter
 * (:68-83), NMS (:84) and the gather of the kept boxes (:85) -- frcnn_topk_desc_nchw_f32,
 * frcnn_decode_clip_filter_nchw_f32, frcnn_nms_sorted_rounds_f32 and frcnn_gather_boxes_f32 enqueued back to back on
 * `stream`, their intermediates in one caller workspace of frcnn_rpn_proposals_workspace_bytes(B, A, k, max_keep)
 * bytes (contents on entry do not matter).  Same results as the four calls; one foreign call and no allocation per
 * step for the host.  Sizes: those frcnn_topk_desc_nchw_supported accepts (A > 8192, max_slices != 1).
 * props_out: (B, max_keep, 4), zero rows after n_keep_out[b]; max_blocks / rounds_used as in
 * frcnn_nms_sorted_rounds_f32 (0 / NULL: every round). */
size_t frcnn_rpn_proposals_workspace_bytes(int B, int A, int k, int max_keep);
int frcnn_rpn_proposals_nchw_f32(const float *const *cls_host, const float *const *reg_host,
                                 const int32_t *hw_host, int n_levels, int na, const float *anchors,
                                 int64_t anchor_batch_stride, int B, int k, float image_h, float image_w,
                                 float min_size, double nms_threshold, int max_keep, float *props_out,
                                 int32_t *n_keep_out, void *workspace, size_t workspace_bytes,
                                 int max_slices, int max_blocks, int32_t *rounds_used, void *stream);

/* ---- K4 NMS ----------------------------------------------------------------------------------
 * torchvision.ops.nms as called at RPN/rpn.py:84 and ROIHead/roi_head.py:215, batched.
 * boxes: (B, max_n, 4), each image's rows ALREADY in candidate order (descending score);
 * n_per_image: (B) int32 device, or NULL for max_n everywhere.  Box j is suppressed by an earlier
 * kept box i when (double)iou_f32(i,j) > iou_threshold; NaN never suppresses.  Stops after
 * max_keep keeps (the reference's [:num_post_nms]).
 * keep_out: (B, max_keep) positions into the image's rows, in keep order; n_keep_out: (B).
 * workspace: frcnn_nms_workspace_bytes(B, max_n) bytes of device memory (0 for max_n <= 2048); its
 * contents on entry do not matter. */
size_t frcnn_nms_workspace_bytes(int B, int max_n);
int frcnn_nms_sorted_f32(const float *boxes, const int32_t *n_per_image, int B, int max_n,
                         double iou_threshold, int max_keep, int32_t *keep_out,
                         int32_t *n_keep_out, void *workspace, size_t workspace_bytes,
                         void *stream);

/* frcnn_nms_sorted_f32 with a bound on the launches (same call sites, RPN/rpn.py:84 / ROIHead/roi_head.py:215;
 * the reference's per-image torchvision call has no such notion).  Above 2048 boxes NMS runs in rounds of 1024
 * candidates, two launches per round; rounds after max_keep is reached exit at once but are still launched,
 * because the host cannot know how many a batch needs (nine of twelve at 12,000 -> 2,000).  max_blocks > 0: the
 * rounds cover at most the first max_blocks 64-candidate blocks of every image (16 per round; the last round possibly
 * only part of its candidates), and the last round's one-workgroup-per-image launch finishes any image that is still
 * not done by itself, 64 candidates at a time: slow when it has much to do, and exact -- keep lists never depend on max_blocks (max_blocks <= 0, or
 * min(max_keep, max_n) > 2048: every round, no tail).  rounds_used: (B) int32, device or host-mapped (pinned) memory,
 * or NULL; written once, by the call's last launch: per image the 64-candidate blocks that had work to do.  A caller
 * feeds max(rounds_used) + slack of one batch to the next call (objectdetection_amd.ops.NmsRounds: slack = 4).
 * LIFETIME: the kernels store into rounds_used when the call's last launch RUNS, not when this function returns: the
 * memory must stay valid (and, if pinned host memory, must not be handed back to an allocator that may reuse it)
 * until that launch has finished on `stream` -- the Python binding keeps its pinned words for the life of the process. */
int frcnn_nms_sorted_rounds_f32(const float *boxes, const int32_t *n_per_image, int B, int max_n,
                                double iou_threshold, int max_keep, int32_t *keep_out,
                                int32_t *n_keep_out, void *workspace, size_t workspace_bytes,
                                int max_blocks, int32_t *rounds_used, void *stream);

/* frcnn_nms_sorted_f32 for callers that cannot allocate device memory.  NO COUNTERPART IN THE REFERENCE: it
 * exists for callers under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same
 * arguments, kernels and results, but the round path's workspace is a file-scope device array inside the
 * library (frcnn_nms_static_scratch_bytes() = 16 MiB, held by every GPU the library runs on until the process
 * ends) instead of a caller buffer; the call allocates and frees nothing.  Sizes with
 * frcnn_nms_workspace_bytes(B, max_n) above that maximum are rejected with FRCNN_EINVAL (B = 16 images of
 * 30,000 boxes or B = 52 of 12,000 fit).  All calls on one GPU share the array: calls that overlap in time
 * on the same GPU (two streams, two host threads) race -- serialise them or use the workspace form. */
size_t frcnn_nms_static_scratch_bytes(void);
int frcnn_nms_sorted_static_f32(const float *boxes, const int32_t *n_per_image, int B, int max_n,
                                double iou_threshold, int max_keep, int32_t *keep_out,
                                int32_t *n_keep_out, void *stream);

/* out[b, j, :] = src[b, idx[b, j], :] for j < count[b], zeros for count[b] <= j < m
 * (proposal[nms_index], RPN/rpn.py:85).  src: (B, n, 4); idx: (B, m); count: (B) or NULL (= m). */
int frcnn_gather_boxes_f32(const float *src, const int32_t *idx, const int32_t *count, int B, int n,
                           int m, float *out, void *stream);

/* ---- K5 IoU / target labelling ---------------------------------------------------------------
 * utils.py:93-120 (box_iou): out (n,m) materialised. */
int frcnn_box_iou_f32(const float *box1, int n, const float *box2, int m, float *out, void *stream);

/* Fused utils.tagging_box (utils.py:261-291) + utils.box_deviation (utils.py:123-171), batched
 * as driven by RPN/rpn.py:105-112 and ROIHead/roi_head.py:99-110.
 * box1: (N,4) shared when box1_batch_stride == 0, else (B,N,4) with that element stride;
 * box2: (B, m_max, 4) with m_per_image (B) valid rows each (NULL = m_max; every image needs >=1);
 * labels: (B, m_max) fp32 class ids or NULL (positives get 1); weights_host: 4 floats or NULL.
 * Row max follows torch.max(dim=1): first maximum wins, the first NaN wins outright.
 * Outputs (each may be NULL): label (B,N) fp32; argmax (B,N) int32; max_iou (B,N); dev (B,N,4). */
int frcnn_iou_label_encode_f32(const float *box1, int64_t box1_batch_stride, const float *box2,
                               const int32_t *m_per_image, const float *labels, int B, int N,
                               int m_max, float neg_threshold, float pos_threshold,
                               const float *weights_host, float *label_out, int32_t *argmax_out,
                               float *max_iou_out, float *dev_out, void *stream);

/* ---- K6 / K7 RoI pooling ---------------------------------------------------------------------
 * rois[(b,j)] = [b, props[b,j,:]] compacted over images (MultiScaleRoIAlign's roi format,
 * ROIHead/roi_pooling.py:51-53).  props: (B,P,4); count: (B) or NULL; rois_out: (B*P,5);
 * total_out: (1) int32 = sum(count). */
int frcnn_build_rois_f32(const float *props, const int32_t *count, int B, int P, float *rois_out,
                         int32_t *total_out, void *stream);

/* torchvision LevelMapper: clamp(floor(4 + log2(sqrt(area)/224) + 1e-6), k_min, k_max) - k_min;
 * NaN -> -1.  rois: (K,5); n_rois: device int32 row count or NULL (= K). */
int frcnn_level_map_f32(const float *rois, const int32_t *n_rois, int K, int k_min, int k_max,
                        int32_t *lvl_out, void *stream);

/* torchvision roi_align (aligned=False) on one level.  feat: (B,C,H,W); rois: (K,5);
 * rows with lvl[r] != lvl_value are skipped when lvl != NULL; rows >= *n_rois are skipped when
 * n_rois != NULL.  out: (K,C,pooled_h,pooled_w) -- untouched rows keep their previous contents
 * (the multi-level wrapper zero-fills first, as torchvision does). */
int frcnn_roi_align_f32(const float *feat, int B, int C, int H, int W, const float *rois,
                        const int32_t *n_rois, int K, int pooled_h, int pooled_w,
                        float spatial_scale, int sampling_ratio, const int32_t *lvl, int lvl_value,
                        float *out, void *stream);

/* Batched form of one MultiScaleRoIAlign level pass over zero-padded proposals, the sync-free
 * path used by RPN -> ROIHead hand-off (ROIHead/roi_head.py:127-130).  props: (B,P,4);
 * count: (B) valid rows per image or NULL (= P).  Output rows are compacted in image order,
 * row(b,j) = sum(count[:b]) + j, exactly the row order of torchvision's concatenated roi list.
 * n_levels == 1: every RoI is pooled from `feat` at `spatial_scale`.  n_levels > 1: a RoI is
 * pooled only when its LevelMapper level (k_min,k_max as in frcnn_level_map_f32) equals `level`;
 * the level-0 call also zero-fills the rows no level takes (torchvision's zeros background).
 * total_out: (1) int32 = sum(count), or NULL.  out: (>= sum(count), C, pooled_h, pooled_w). */
int frcnn_roi_align_batched_f32(const float *feat, int B, int C, int H, int W, const float *props,
                                const int32_t *count, int P, int pooled_h, int pooled_w,
                                float spatial_scale, int sampling_ratio, int level, int n_levels,
                                int k_min, int k_max, float *out, int32_t *total_out, void *stream);

/* Every level of MultiScaleRoIAlign in one call (ROIHead/roi_pooling.py:44-53, call site
 * ROIHead/roi_head.py:127-130, on the pyramid of BackBone/fpn.py:91-96): what n_levels calls of
 * frcnn_roi_align_batched_f32 with level = 0 .. n_levels-1 produce, same row order, same zero rows for the RoIs no
 * level takes.  feats: HOST array of n_levels device pointers (B,C,h_l,w_l); hw: HOST array (h_0,w_0,h_1,w_1,...);
 * scales: HOST array of spatial scales; n_levels <= 6.  For 7x7 bins at sampling_ratio 2 the levels are pooled by
 * ONE launch (tiles of each level in LDS, 16 channels per item; csrc/roi_tile.hip) plus one small launch per level
 * of several tiles for RoIs wider than 105 cells; a single map small enough for the plane kernel, and every other
 * bin count or sampling ratio, goes level by level through frcnn_roi_align_batched_f32. */
int frcnn_roi_align_levels_f32(const float *const *feats, const int32_t *hw, const float *scales, int n_levels,
                               int B, int C, const float *props, const int32_t *count, int P, int pooled_h,
                               int pooled_w, int sampling_ratio, int k_min, int k_max, float *out,
                               int32_t *total_out, void *stream);

/* The same pooling with LANES = CHANNELS on channel-innermost maps (csrc/roi_cl.hip; same reference lines:
 * ROIHead/roi_pooling.py:44-53, ROIHead/roi_head.py:127-130, BackBone/fpn.py:91-96).  A wave takes (RoI, 64
 * channels), keeps the 49 sums of a channel in registers and contracts every cell row of a RoI's footprint once: the
 * row comes through a wave-private LDS ring (buffer_load_dwordx4 ... lds) when the RoI's x taps span <= 16 cells, as
 * coalesced 256-byte loads into registers otherwise; no window of the map is staged, no RoI size limit.
 * maps_channels_last != 0: feats[l] is laid out (B, h_l, w_l, C) (torch.channels_last memory) and read in place;
 * maps_channels_last == 0: feats[l] is (B, C, h_l, w_l) and is first copied to that layout inside `workspace`.
 * Same outputs as frcnn_roi_align_levels_f32 (row order, zero rows, total_out), values equal up to the summation
 * order.  Needs 7 x 7 bins, sampling_ratio 2, C % 64 == 0, P < 2^24, h_l * w_l * C * 4 < 2^31 (FRCNN_EINVAL
 * otherwise); workspace: frcnn_roi_align_levels_cl_workspace_bytes(...) bytes (0 = unsupported shape), 256-byte
 * aligned, caller-owned, contents need not survive between calls. */
size_t frcnn_roi_align_levels_cl_workspace_bytes(const int32_t *hw, int n_levels, int B, int C, int P,
                                                 int maps_channels_last);
int frcnn_roi_align_levels_cl_f32(const float *const *feats, const int32_t *hw, const float *scales, int n_levels,
                                  int B, int C, int maps_channels_last, const float *props, const int32_t *count,
                                  int P, int pooled_h, int pooled_w, int sampling_ratio, int k_min, int k_max,
                                  float *out, int32_t *total_out, void *workspace, size_t workspace_bytes,
                                  void *stream);

/* Backward of frcnn_roi_align_batched_f32 with respect to `feat`: what torch autograd runs through
 * torchvision's MultiScaleRoIAlign at ROIHead/roi_pooling.py:51-53 / ROIHead/roi_head.py:127-130 (the reference's
 * RoI losses train the backbone through the pooling).  Same arguments as the forward pass of one level;
 * grad_out: (>= sum(count), C, pooled_h, pooled_w) in the forward's row order; grad_feat: (B, C, H, W), written
 * in full (zeros where no RoI of this level reaches).  Every sample sends grad * w / (sampling_ratio^2) to its
 * four taps (torchvision roi_align_backward); boxes get no gradient.  Planes that fit LDS are summed in RoI order by
 * one workgroup per (image, channel chunk): bit-identical from run to run; larger planes and small batches add partial
 * sums with global float atomics, whose order is not fixed (as in torchvision).  grad_out must be finite: the 7 x 7
 * products run as dense matrix products, so one Inf / NaN of a RoI's gradient reaches every cell of that RoI's footprint
 * (torchvision: only the cells the bin's samples touch) and up to three cells beside it in the same rows. */
int frcnn_roi_align_batched_backward_f32(const float *grad_out, int B, int C, int H, int W,
                                         const float *props, const int32_t *count, int P, int pooled_h,
                                         int pooled_w, float spatial_scale, int sampling_ratio, int level,
                                         int n_levels, int k_min, int k_max, float *grad_feat, void *stream);

/* torchvision roi_pool (max) -- north_star extra, not called by the reference.
 * out: (K,C,ph,pw) fp32; argmax: (K,C,ph,pw) int32 flat index h*W+w, -1 for an empty bin. */
int frcnn_roi_pool_f32(const float *feat, int B, int C, int H, int W, const float *rois, int K,
                       int pooled_h, int pooled_w, float spatial_scale, float *out, int32_t *argmax,
                       void *stream);

/* The same max pooling over zero-padded proposals, batched like frcnn_roi_align_batched_f32 (props: (B,P,4);
 * count: (B) or NULL; output rows compacted in image order; total_out: (1) int32 or NULL): the form that runs
 * at the headline shape (16 x 2,000 RoIs, C = 512) with the planes resident in LDS.  7 x 7 bins and a plane
 * that fits LDS four channels at a time only (FRCNN_EINVAL otherwise: use frcnn_roi_pool_f32).  Not called by
 * the reference (north_star: "RoI pooling to 7x7 ... argmax"); semantics are torchvision roi_pool's. */
int frcnn_roi_pool_batched_f32(const float *feat, int B, int C, int H, int W, const float *props,
                               const int32_t *count, int P, int pooled_h, int pooled_w, float spatial_scale,
                               float *out, int32_t *argmax, int32_t *total_out, void *stream);

/* ---- ROIHead eval post-process ---------------------------------------------
 * ROIHead/roi_head.py:178-209 for a whole batch in one launch.  props: (N,4) the images' proposal
 * lists concatenated; regs: (N,K,4); scores: (N,K) (softmax output); row_off: (B+1) int32 device,
 * rows of image b are [row_off[b], row_off[b+1]), at most max_rows of them; image_sizes: (B,2)
 * fp32 device (h, w).  Candidate (j, c) of an image is decoded (box_regression), clipped to that
 * image's size and kept when score > score_thresh, w > min_size, h > min_size and c > 0, in the
 * reference's flat order j*K + c.  Outputs are (B, max_rows*K[,4]) with the first count_out[b]
 * rows of each image valid: boxes, scores, labels (int32 class ids); rows beyond the count are
 * unspecified (labels_out's are used as scratch between the two launches). */
int frcnn_roi_detect_filter_f32(const float *props, const float *regs, const float *scores,
                                const int32_t *row_off, const float *image_sizes, int B, int K,
                                int max_rows, float score_thresh, float min_size, float *boxes_out,
                                float *scores_out, int32_t *labels_out, int32_t *count_out, void *stream);

/* ---- sampled-gather + losses ---------------------------------------------
 * RPN/rpn.py:122-137 in one launch: rows `index[s]` (flat b*A+a, S of them, negatives first) are
 * gathered from scores (rows), regs (rows,4), labels (rows), reg_labels (rows,4);
 * loss_out[0] = mean BCE-with-logits, loss_out[1] = utils.smooth_l1_loss (utils.py:315-336) with
 * `beta`, mean over all 4*S elements (quirk Q7: negatives included).  grad_scores (rows) and
 * grad_regs (rows,4), when not NULL, are zero-filled and receive d loss_out[0] / d scores and
 * d loss_out[1] / d regs.  An index outside [0, rows) turns both losses into NaN. */
int frcnn_rpn_sampled_loss_f32(const float *scores, const float *regs, const float *labels,
                               const float *reg_labels, const int64_t *index, int S, int64_t rows,
                               float beta, float *loss_out, float *grad_scores, float *grad_regs,
                               void *stream);

/* ROIHead/roi_head.py:141-152 in one launch.  logits: (N,Kc); pred: (N,Kc,4); cls_label: (N) fp32
 * class ids (0 = background, truncated like .long()); reg_label: (N,4).
 * loss_out[0] = mean cross-entropy; loss_out[1] = smooth-L1 mean over the 4*n_pos elements of
 * pred[n, label_n] - reg_label[n] for rows with label > 0 (NaN when n_pos == 0, as torch's mean of
 * an empty tensor).  grad_logits (N,Kc), grad_pred (N,Kc,4): written in full when not NULL.
 * A label outside [0, Kc) turns both losses into NaN (torch raises). */
int frcnn_roi_loss_f32(const float *logits, const float *pred, const float *cls_label,
                       const float *reg_label, int N, int Kc, float beta, float *loss_out,
                       float *grad_logits, float *grad_pred, void *stream);

/* ---- batch assembly in front of the path ----------------------------------
 * generalized_transform.py:58-64 + utils.py:362-369 for ONE image: bilinear resize (torch
 * F.interpolate, align_corners=False, scale = in/out as with recompute_scale_factor=True) of
 * src (C,in_h,in_w) to (out_h,out_w), written into the top-left corner of dst (C,canvas_h,canvas_w),
 * zeros elsewhere.  out == in gives the plain zero-padded copy of utils.batched_images. */
int frcnn_resize_pad_f32(const float *src, int C, int in_h, int in_w, int out_h, int out_w,
                         float *dst, int canvas_h, int canvas_w, void *stream);

/* The same for a whole batch in one launch (per 32 images): src_host is a HOST array of n device pointers to
 * (C,in_h,in_w) images, shape_host a HOST array (n,4) of (in_h, in_w, out_h, out_w); dst is the
 * (n, C, canvas_h, canvas_w) batch.  Bit-identical to n calls of frcnn_resize_pad_f32. */
int frcnn_resize_pad_batch_f32(const float *const *src_host, const int32_t *shape_host, int n, int C,
                               float *dst, int canvas_h, int canvas_w, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FRCNN_HIP_H */
