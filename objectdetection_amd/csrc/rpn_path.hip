// rpn_path.hip -- the RPN proposal half (RPN/rpn.py:54-85) behind ONE C call: top-k over the head's NCHW scores,
// decode + clip + small-box filter, NMS, gather of the kept boxes -- the four stage entry points of this library
// enqueued back to back on the caller's stream, their intermediates carved out of one caller workspace.  Nothing
// here launches a kernel of its own; the point is the host side: one foreign call and no allocation per step
// instead of four calls and ten tensors (a batch of one image is bound by exactly that: 0.33 ms of Python per
// step against ~0.2 ms of GPU work).
#include "common.h"

static size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct RpnPathWs {
    size_t topk, top_val, top_idx, boxes, scores, src, count, keep, nms, total;
};

static RpnPathWs rpn_path_plan(int B, int A, int k, int max_keep)
{
    RpnPathWs w;
    size_t off = 0;
    const size_t Bk = (size_t)B * (size_t)k;
    w.topk = off; off += up256(frcnn_topk_workspace_bytes(B, A, k));
    w.top_val = off; off += up256(Bk * 4);
    w.top_idx = off; off += up256(Bk * 4);
    w.boxes = off; off += up256(Bk * 16);
    w.scores = off; off += up256(Bk * 4);
    w.src = off; off += up256(Bk * 4);
    w.count = off; off += up256((size_t)B * 4);
    w.keep = off; off += up256((size_t)B * (size_t)(max_keep > 0 ? max_keep : 1) * 4);
    w.nms = off; off += up256(frcnn_nms_workspace_bytes(B, k));
    w.total = off;
    return w;
}

extern "C" size_t frcnn_rpn_proposals_workspace_bytes(int B, int A, int k, int max_keep)
{
    if (B <= 0 || A <= 0 || k <= 0 || k > A || max_keep < 0) return 0;
    return rpn_path_plan(B, A, k, max_keep).total;
}

extern "C" int frcnn_rpn_proposals_nchw_f32(const float *const *cls_host, const float *const *reg_host,
                                            const int32_t *hw_host, int n_levels, int na, const float *anchors,
                                            int64_t anchor_batch_stride, int B, int k, float image_h, float image_w,
                                            float min_size, double nms_threshold, int max_keep, float *props_out,
                                            int32_t *n_keep_out, void *workspace, size_t workspace_bytes,
                                            int max_slices, int max_blocks, int32_t *rounds_used, void *stream)
{
    FRCNN_REQUIRE(cls_host && reg_host && hw_host && anchors && B > 0 && k > 0 && max_keep > 0,
                  "rpn_proposals: bad arguments");
    FRCNN_REQUIRE(n_levels >= 1 && n_levels <= FRCNN_MAX_LEVELS && na > 0, "rpn_proposals: bad level table");
    long long A = 0;
    for (int l = 0; l < n_levels; ++l) {
        FRCNN_REQUIRE(hw_host[2 * l] > 0 && hw_host[2 * l + 1] > 0, "rpn_proposals: bad level %d", l);
        A += (long long)hw_host[2 * l] * hw_host[2 * l + 1] * na;
    }
    FRCNN_REQUIRE(A < (1ll << 31) && k <= A, "rpn_proposals: k (%d) > A (%lld) or A too large", k, A);
    FRCNN_REQUIRE(frcnn_topk_desc_nchw_supported(B, (int)A, k, max_slices),
                  "rpn_proposals: A=%lld is outside the in-place top-k's range (use the stage entry points)", A);
    FRCNN_REQUIRE(props_out && n_keep_out && workspace, "rpn_proposals: null pointer");
    const RpnPathWs w = rpn_path_plan(B, (int)A, k, max_keep);
    if (workspace_bytes < w.total) {
        frcnn_set_error("rpn_proposals: workspace too small (%zu < %zu)", workspace_bytes, w.total);
        return FRCNN_EWORKSPACE;
    }
    char *base = (char *)workspace;
    float *top_val = (float *)(base + w.top_val);
    int32_t *top_idx = (int32_t *)(base + w.top_idx);
    float *boxes = (float *)(base + w.boxes);
    float *scores = (float *)(base + w.scores);
    int32_t *src = (int32_t *)(base + w.src);
    int32_t *count = (int32_t *)(base + w.count);
    int32_t *keep = (int32_t *)(base + w.keep);
    int rc = frcnn_topk_desc_nchw_f32(cls_host, hw_host, n_levels, na, B, k, top_val, top_idx, base + w.topk,
                                      w.top_val - w.topk, max_slices, stream);                       // rpn.py:65-67
    if (rc != FRCNN_OK) return rc;
    rc = frcnn_decode_clip_filter_nchw_f32(anchors, anchor_batch_stride, reg_host, hw_host, n_levels, na, top_idx,
                                           top_val, B, k, image_h, image_w, min_size, boxes, scores, src, count,
                                           stream);                                                  // rpn.py:68-83
    if (rc != FRCNN_OK) return rc;
    rc = frcnn_nms_sorted_rounds_f32(boxes, count, B, k, nms_threshold, max_keep, keep, n_keep_out, base + w.nms,
                                     w.total - w.nms, max_blocks, rounds_used, stream);              // rpn.py:84
    if (rc != FRCNN_OK) return rc;
    return frcnn_gather_boxes_f32(boxes, keep, n_keep_out, B, k, max_keep, props_out, stream);       // rpn.py:85
}
