/*
 * frcnn_oracle.c -- CPU restatement of the Faster R-CNN proposal / RoI hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load the library built from this file; the
 * product path (objectdetection_amd/) never imports it and fails loudly when its HIP
 * extension is missing.
 *
 * Every function restates, operation by operation and in the same fp32 association
 * order, one function of the reference (paths relative to /root/reference/):
 *   code/FasterRCNN/utils.py, code/FasterRCNN/RPN/anchor_generator.py,
 *   code/FasterRCNN/RPN/rpn.py, code/FasterRCNN/ROIHead/roi_pooling.py
 * or, for the two third-party ops the reference calls but does not vendor
 * (torchvision.ops.nms, torchvision.ops.MultiScaleRoIAlign -- torchvision is NOT
 * installed in this image and its version is unpinned by the reference), the published
 * algorithm as recorded in SURVEY.md section 8(c).
 *
 * Pinning status:
 *   - anchors, box_iou, tagging_box, box_deviation, box_regression, box_clip,
 *     remove_small_box: pinned against golden vectors captured by importing the
 *     reference itself (tests/golden/make_golden.py -> the .npz files in tests/golden/).
 *   - nms, roi_align / level mapper, roi_pool, top-k tie rule: PARITY UNPINNED -- the
 *     reference holds no fixture for them and torchvision cannot be imported here.
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -fno-fast-math -fopenmp (oracle/Makefile).
 * No FMA contraction, no reassociation: x86-64 SSE2 float arithmetic is IEEE fp32.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }
int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------
 * Anchors.  RPN/anchor_generator.py:43-64 (get_base_anchor) and :66-95 (generate_anchor)
 * ---------------------------------------------------------------------------------- */

/* One (size, ratio) pair per base anchor: h = size, w = size * ratio (fp32 multiply),
 * anchor = [-w, -h, w, h] / 2.   anchor_generator.py:59-61 */
void orc_base_anchors(const float *size, const float *ratio, int na, float *out)
{
    for (int a = 0; a < na; ++a) {
        float h = size[a];
        float w = size[a] * ratio[a];
        out[a * 4 + 0] = -w / 2.0f;
        out[a * 4 + 1] = -h / 2.0f;
        out[a * 4 + 2] = w / 2.0f;
        out[a * 4 + 3] = h / 2.0f;
    }
}

/* grid_x = arange(fw) * stride_x, grid_y = arange(fh) * stride_y (fp32), row-major cells
 * (y outer, x inner), anchor = [gx, gy, gx, gy] + base.   anchor_generator.py:86-93
 * out: (fh*fw, na, 4) */
void orc_anchor_grid(int fh, int fw, float stride_y, float stride_x, const float *base, int na,
                     float *out)
{
    for (int y = 0; y < fh; ++y) {
        float gy = (float)y * stride_y;
        for (int x = 0; x < fw; ++x) {
            float gx = (float)x * stride_x;
            float *o = out + ((size_t)(y * fw + x) * na) * 4;
            for (int a = 0; a < na; ++a) {
                o[a * 4 + 0] = gx + base[a * 4 + 0];
                o[a * 4 + 1] = gy + base[a * 4 + 1];
                o[a * 4 + 2] = gx + base[a * 4 + 2];
                o[a * 4 + 3] = gy + base[a * 4 + 3];
            }
        }
    }
}

/* ------------------------------------------------------------------------------------
 * Box math.  utils.py
 * ---------------------------------------------------------------------------------- */

/* utils.py:93-120.  iou = inter / ((area1 + area2) - inter); 0/0 -> NaN, no epsilon. */
static inline float iou_pair(const float *a, float area_a, const float *b, float area_b)
{
    float ltx = a[0] > b[0] ? a[0] : b[0]; /* torch.max: NaN handling irrelevant for finite boxes */
    float lty = a[1] > b[1] ? a[1] : b[1];
    float rbx = a[2] < b[2] ? a[2] : b[2];
    float rby = a[3] < b[3] ? a[3] : b[3];
    float jw = rbx - ltx;
    float jh = rby - lty;
    jw = jw < 0.0f ? 0.0f : jw; /* clamp(min=0) */
    jh = jh < 0.0f ? 0.0f : jh;
    float inter = jw * jh;
    return inter / ((area_a + area_b) - inter);
}

static inline float box_area(const float *b)
{
    float w = b[2] - b[0];
    float h = b[3] - b[1];
    return w * h;
}

/* utils.py:93-120 -> (n, m) matrix */
void orc_box_iou(const float *box1, int n, const float *box2, int m, float *out)
{
    float *area2 = (float *)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1));
    for (int j = 0; j < m; ++j) area2[j] = box_area(box2 + 4 * j);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int i = 0; i < n; ++i) {
        float a1 = box_area(box1 + 4 * i);
        for (int j = 0; j < m; ++j)
            out[(size_t)i * m + j] = iou_pair(box1 + 4 * i, a1, box2 + 4 * j, area2[j]);
    }
    free(area2);
}

/* Row max/argmax the way torch.max(dim=1) reduces on CPU: a later value replaces the
 * running max only when NOT (value <= max); the first NaN wins and stops the scan.
 * (utils.py:136, utils.py:277) */
static inline void row_max(const float *box1_row, float a1, const float *box2, const float *area2,
                           int m, float *max_out, int *arg_out)
{
    float best = iou_pair(box1_row, a1, box2, area2[0]);
    int arg = 0;
    if (!(best != best)) {
        for (int j = 1; j < m; ++j) {
            float v = iou_pair(box1_row, a1, box2 + 4 * j, area2[j]);
            if (!(v <= best)) {
                best = v;
                arg = j;
                if (v != v) break;
            }
        }
    }
    *max_out = best;
    *arg_out = arg;
}

/* Fused restatement of utils.tagging_box (utils.py:261-291) and utils.box_deviation
 * (utils.py:123-171) for one image.  labels may be NULL (then positives get 1).
 * weights: 4 floats (ones when the reference passes None).
 * Outputs (any may be NULL): label (n) fp32, argmax (n) int64, max_iou (n), dev (n,4). */
void orc_iou_label_encode(const float *box1, int n, const float *box2, int m, float neg_thr,
                          float pos_thr, const float *labels, const float *weights, float *label_out,
                          int64_t *argmax_out, float *maxiou_out, float *dev_out)
{
    float *area2 = (float *)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1));
    for (int j = 0; j < m; ++j) area2[j] = box_area(box2 + 4 * j);
    const float one[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    const float *wt = weights ? weights : one;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int i = 0; i < n; ++i) {
        const float *b1 = box1 + 4 * i;
        float a1 = box_area(b1);
        float mx;
        int arg;
        row_max(b1, a1, box2, area2, m, &mx, &arg);
        if (maxiou_out) maxiou_out[i] = mx;
        if (argmax_out) argmax_out[i] = arg;
        if (label_out) {
            /* utils.py:275-290: zeros; pos -> 1 / labels[arg]; then neg -> 0; then mid -> -1 */
            float lab = 0.0f;
            if (mx > pos_thr) lab = labels ? labels[arg] : 1.0f;
            if (mx < neg_thr) lab = 0.0f;
            if (mx >= neg_thr && mx <= pos_thr) lab = -1.0f;
            label_out[i] = lab;
        }
        if (dev_out) {
            /* utils.py:147-161 */
            const float *b2 = box2 + 4 * arg;
            float w1 = b1[2] - b1[0], h1 = b1[3] - b1[1];
            float cx1 = b1[0] + w1 / 2.0f, cy1 = b1[1] + h1 / 2.0f;
            float w2 = b2[2] - b2[0], h2 = b2[3] - b2[1];
            float cx2 = b2[0] + w2 / 2.0f, cy2 = b2[1] + h2 / 2.0f;
            dev_out[4 * i + 0] = wt[0] * ((cx2 - cx1) / w1);
            dev_out[4 * i + 1] = wt[1] * ((cy2 - cy1) / h1);
            dev_out[4 * i + 2] = wt[2] * logf(w2 / w1);
            dev_out[4 * i + 3] = wt[3] * logf(h2 / h1);
        }
    }
    free(area2);
}

/* utils.py:173-208.  box (n,4), regs (n,4K) laid out [dx,dy,dw,dh] x K, out (n,4K)
 * laid out [x1,y1,x2,y2] x K.  NOTE (quirk Q1): centre = (w/2, h/2), the box origin
 * is NOT added (utils.py:188-189). */
void orc_box_regression(const float *box, const float *regs, int n, int K, float *out)
{
    const float clampv = (float)log(1000.0 / 16.0); /* utils.py:193 */
    for (int i = 0; i < n; ++i) {
        float bh = box[4 * i + 3] - box[4 * i + 1];
        float bw = box[4 * i + 2] - box[4 * i + 0];
        float cx = bw / 2.0f;
        float cy = bh / 2.0f;
        for (int k = 0; k < K; ++k) {
            const float *r = regs + (size_t)i * 4 * K + 4 * k;
            float dx = r[0], dy = r[1], dw = r[2], dh = r[3];
            dw = dw > clampv ? clampv : dw; /* torch.clamp(max=): NaN propagates */
            dh = dh > clampv ? clampv : dh;
            float px = cx + dx * bw;
            float py = cy + dy * bh;
            float ph = expf(dh) * bh;
            float pw = expf(dw) * bw;
            float *o = out + (size_t)i * 4 * K + 4 * k;
            o[0] = px - pw / 2.0f;
            o[1] = py - ph / 2.0f;
            o[2] = px + pw / 2.0f;
            o[3] = py + ph / 2.0f;
        }
    }
}

/* Backward of utils.py:173-208 as torch autograd evaluates it (grad of each forward op in turn).
 * grad_out (n,4K) -> grad_regs (n,4K), grad_box (n,4) (may be NULL).  torch.clamp(max=) passes the
 * gradient where the input <= max. */
void orc_box_regression_backward(const float *box, const float *regs, const float *grad_out, int n,
                                 int K, float *grad_regs, float *grad_box)
{
    const float clampv = (float)log(1000.0 / 16.0);
    for (int i = 0; i < n; ++i) {
        float bw = box[4 * i + 2] - box[4 * i + 0];
        float bh = box[4 * i + 3] - box[4 * i + 1];
        float gbw = 0.0f, gbh = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float *r = regs + (size_t)i * 4 * K + 4 * k;
            const float *g = grad_out + (size_t)i * 4 * K + 4 * k;
            float *o = grad_regs + (size_t)i * 4 * K + 4 * k;
            float dwc = r[2] > clampv ? clampv : r[2];
            float dhc = r[3] > clampv ? clampv : r[3];
            float ew = expf(dwc), eh = expf(dhc);
            float gpx = g[0] + g[2], gpy = g[1] + g[3];
            float gpw = g[2] / 2.0f - g[0] / 2.0f, gph = g[3] / 2.0f - g[1] / 2.0f;
            o[0] = gpx * bw;
            o[1] = gpy * bh;
            o[2] = r[2] <= clampv ? (gpw * bw) * ew : 0.0f;
            o[3] = r[3] <= clampv ? (gph * bh) * eh : 0.0f;
            gbw += gpx / 2.0f + gpx * r[0] + gpw * ew;
            gbh += gpy / 2.0f + gpy * r[1] + gph * eh;
        }
        if (grad_box) {
            grad_box[4 * i + 0] = -gbw;
            grad_box[4 * i + 1] = -gbh;
            grad_box[4 * i + 2] = gbw;
            grad_box[4 * i + 3] = gbh;
        }
    }
}

/* generalized_transform.py:58-64 (torch upsample_bilinear2d, align_corners=False, scale = in/out)
 * + utils.py:362-369 (copy into a zero canvas) for one image. */
void orc_resize_pad(const float *src, int C, int in_h, int in_w, int out_h, int out_w, float *dst,
                    int canvas_h, int canvas_w)
{
    const float scale_h = (float)in_h / (float)out_h, scale_w = (float)in_w / (float)out_w;
    memset(dst, 0, sizeof(float) * (size_t)C * canvas_h * canvas_w);
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < out_h; ++y) {
            float sy = scale_h * ((float)y + 0.5f) - 0.5f;
            sy = sy < 0.0f ? 0.0f : sy;
            int y0 = (int)sy;
            y0 = y0 > in_h - 1 ? in_h - 1 : y0;
            int y1 = y0 < in_h - 1 ? y0 + 1 : y0;
            float ly = sy - (float)y0, hy = 1.0f - ly;
            for (int x = 0; x < out_w; ++x) {
                float sx = scale_w * ((float)x + 0.5f) - 0.5f;
                sx = sx < 0.0f ? 0.0f : sx;
                int x0 = (int)sx;
                x0 = x0 > in_w - 1 ? in_w - 1 : x0;
                int x1 = x0 < in_w - 1 ? x0 + 1 : x0;
                float lx = sx - (float)x0, hx = 1.0f - lx;
                const float *p = src + (size_t)c * in_h * in_w;
                dst[((size_t)c * canvas_h + y) * canvas_w + x] =
                    hy * (hx * p[y0 * in_w + x0] + lx * p[y0 * in_w + x1]) +
                    ly * (hx * p[y1 * in_w + x0] + lx * p[y1 * in_w + x1]);
            }
        }
}

/* utils.py:331-333 element: x = |d|; x < beta ? 0.5 x^2 / beta : x - 0.5 beta; *g = d/d y_hat. */
static float orc_sl1(float y_hat, float y, float beta, float *g)
{
    float d = y_hat - y, x = fabsf(d);
    if (x < beta) {
        *g = d / beta;
        return 0.5f * x * x / beta;
    }
    *g = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    return x - 0.5f * beta;
}

/* RPN/rpn.py:122-137: gather rows index[s] and reduce (sums in double, then fp32 means).
 * loss[0] = F.binary_cross_entropy_with_logits mean, loss[1] = utils.smooth_l1_loss mean over 4*S
 * elements.  grad_scores (rows) / grad_regs (rows*4) are zero-filled then accumulated. */
void orc_rpn_sampled_loss(const float *scores, const float *regs, const float *labels,
                          const float *reg_labels, const int64_t *index, int S, int64_t rows, float beta,
                          float *loss, float *grad_scores, float *grad_regs)
{
    double cls = 0.0, reg = 0.0;
    memset(grad_scores, 0, sizeof(float) * (size_t)rows);
    memset(grad_regs, 0, sizeof(float) * 4 * (size_t)rows);
    for (int s = 0; s < S; ++s) {
        int64_t i = index[s];
        float x = scores[i], y = labels[i];
        cls += (double)((x > 0.0f ? x : 0.0f) - x * y + log1pf(expf(-fabsf(x))));
        grad_scores[i] += (1.0f / (1.0f + expf(-x)) - y) / (float)S;
        for (int c = 0; c < 4; ++c) {
            float g;
            reg += (double)orc_sl1(regs[4 * i + c], reg_labels[4 * i + c], beta, &g);
            grad_regs[4 * i + c] += g / (4.0f * (float)S);
        }
    }
    loss[0] = (float)(cls / S);
    loss[1] = (float)(reg / (4.0 * S));
}

/* ROIHead/roi_head.py:141-152: F.cross_entropy mean + smooth-L1 of the positives' per-class rows. */
void orc_roi_loss(const float *logits, const float *pred, const float *cls_label, const float *reg_label,
                  int N, int Kc, float beta, float *loss, float *grad_logits, float *grad_pred)
{
    double ce = 0.0, reg = 0.0;
    int npos = 0;
    for (int n = 0; n < N; ++n) npos += cls_label[n] > 0.0f;
    memset(grad_pred, 0, sizeof(float) * 4 * (size_t)N * Kc);
    for (int n = 0; n < N; ++n) {
        const float *x = logits + (size_t)n * Kc;
        int li = (int)cls_label[n];
        float m = x[0];
        for (int k = 1; k < Kc; ++k) m = x[k] > m ? x[k] : m;
        float sum = 0.0f;
        for (int k = 0; k < Kc; ++k) sum += expf(x[k] - m);
        float lse = m + logf(sum);
        ce += (double)(lse - x[li]);
        for (int k = 0; k < Kc; ++k)
            grad_logits[(size_t)n * Kc + k] = (expf(x[k] - lse) - (k == li ? 1.0f : 0.0f)) / (float)N;
        if (cls_label[n] > 0.0f) {
            for (int c = 0; c < 4; ++c) {
                float g;
                size_t e = ((size_t)n * Kc + li) * 4 + c;
                reg += (double)orc_sl1(pred[e], reg_label[4 * n + c], beta, &g);
                grad_pred[e] = g / (4.0f * (float)npos);
            }
        }
    }
    loss[0] = (float)(ce / N);
    loss[1] = npos ? (float)(reg / (4.0 * npos)) : NAN;
}

/* utils.py:242-259: x in clamp[0, image_w], y in clamp[0, image_h] (inclusive bounds). */
void orc_box_clip(const float *box, int n, float image_h, float image_w, float *out)
{
    for (int i = 0; i < n; ++i) {
        for (int c = 0; c < 4; ++c) {
            float hi = (c & 1) ? image_h : image_w;
            float v = box[4 * i + c];
            v = v < 0.0f ? 0.0f : v;
            v = v > hi ? hi : v;
            out[4 * i + c] = v;
        }
    }
}

/* utils.py:231-239: keep rows with (w > t) AND (h > t), ascending index.  Returns count. */
int orc_remove_small_box(const float *box, int n, float size_thresh, int64_t *index_out)
{
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        float w = box[4 * i + 2] - box[4 * i + 0];
        float h = box[4 * i + 3] - box[4 * i + 1];
        if (w > size_thresh && h > size_thresh) index_out[cnt++] = i;
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------
 * Ordering rule used for top-k and for NMS's score sort (PARITY UNPINNED for ties:
 * torch.topk's tie order is implementation-defined, see SURVEY.md section 7 hard part 2).
 * Definition adopted by this build: descending by an order-preserving unsigned key of
 * the fp32 score (any NaN ranks above +inf, -0.0 below +0.0), ties by ascending index.
 * ---------------------------------------------------------------------------------- */
static inline uint32_t score_key(float s)
{
    uint32_t u;
    memcpy(&u, &s, 4);
    if (s != s) return 0xFFFFFFFFu;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

typedef struct {
    uint32_t key;
    int32_t idx;
} keyidx_t;

static int cmp_keyidx(const void *pa, const void *pb)
{
    const keyidx_t *a = (const keyidx_t *)pa, *b = (const keyidx_t *)pb;
    if (a->key != b->key) return a->key > b->key ? -1 : 1;
    return a->idx < b->idx ? -1 : (a->idx > b->idx ? 1 : 0);
}

/* order_out (n) = permutation sorting scores descending under the rule above. */
void orc_argsort_desc(const float *scores, int n, int64_t *order_out)
{
    keyidx_t *v = (keyidx_t *)malloc(sizeof(keyidx_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        v[i].key = score_key(scores[i]);
        v[i].idx = i;
    }
    qsort(v, (size_t)n, sizeof(keyidx_t), cmp_keyidx);
    for (int i = 0; i < n; ++i) order_out[i] = v[i].idx;
    free(v);
}

/* RPN/rpn.py:65-66: k = min(num_pre_nms, A); _, idx = torch.topk(score, k) (sorted). */
void orc_topk(const float *scores, int n, int k, int64_t *idx_out, float *val_out)
{
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    orc_argsort_desc(scores, n, order);
    if (k > n) k = n;
    for (int i = 0; i < k; ++i) {
        idx_out[i] = order[i];
        if (val_out) val_out[i] = scores[order[i]];
    }
    free(order);
}

/* ------------------------------------------------------------------------------------
 * torchvision.ops.nms(boxes, scores, iou_threshold) -- call sites RPN/rpn.py:84 and
 * ROIHead/roi_head.py:215.  Greedy; candidate order = descending score (rule above);
 * box j is suppressed by an earlier kept box i when (double)ovr > iou_threshold with
 * ovr = inter / ((area_i + area_j) - inter) in fp32; NaN ovr never suppresses.
 * Returns the number of kept boxes; keep_out holds ORIGINAL indices in keep order.
 * max_keep < 0 means unlimited (the reference slices [:num_post_nms] afterwards, which
 * equals stopping early).
 * ---------------------------------------------------------------------------------- */
int orc_nms(const float *boxes, const float *scores, int n, double iou_threshold, int max_keep,
            int64_t *keep_out)
{
    if (n <= 0) return 0;
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    float *areas = (float *)malloc(sizeof(float) * (size_t)n);
    unsigned char *sup = (unsigned char *)calloc((size_t)n, 1);
    orc_argsort_desc(scores, n, order);
    for (int i = 0; i < n; ++i) areas[i] = box_area(boxes + 4 * i);
    int nk = 0;
    for (int _i = 0; _i < n; ++_i) {
        if (max_keep >= 0 && nk >= max_keep) break;
        int64_t i = order[_i];
        if (sup[i]) continue;
        keep_out[nk++] = i;
        const float *bi = boxes + 4 * i;
        float ia = areas[i];
        for (int _j = _i + 1; _j < n; ++_j) {
            int64_t j = order[_j];
            if (sup[j]) continue;
            const float *bj = boxes + 4 * j;
            float xx1 = bi[0] > bj[0] ? bi[0] : bj[0];
            float yy1 = bi[1] > bj[1] ? bi[1] : bj[1];
            float xx2 = bi[2] < bj[2] ? bi[2] : bj[2];
            float yy2 = bi[3] < bj[3] ? bi[3] : bj[3];
            float w = xx2 - xx1;
            float h = yy2 - yy1;
            w = w > 0.0f ? w : 0.0f;
            h = h > 0.0f ? h : 0.0f;
            float inter = w * h;
            float ovr = inter / ((ia + areas[j]) - inter);
            if ((double)ovr > iou_threshold) sup[j] = 1;
        }
    }
    free(order);
    free(areas);
    free(sup);
    return nk;
}

/* ------------------------------------------------------------------------------------
 * torchvision.ops.MultiScaleRoIAlign -- call site ROIHead/roi_pooling.py:44-53.
 * ---------------------------------------------------------------------------------- */

/* Level mapper: lvl = floor(4 + log2(sqrt(area)/224) + 1e-6), clamped as
 * torch.clamp(min=k_min, max=k_max) (min applied first, then max), minus k_min.
 * A NaN level (negative area) maps to -1 = "no level" (the row stays zero). */
void orc_level_map(const float *boxes, int n, int k_min, int k_max, int32_t *lvl_out)
{
    for (int i = 0; i < n; ++i) {
        float area = (boxes[4 * i + 2] - boxes[4 * i + 0]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
        float s = sqrtf(area);
        float t = floorf((4.0f + log2f(s / 224.0f)) + 1e-6f);
        if (t != t) {
            lvl_out[i] = -1;
            continue;
        }
        t = t < (float)k_min ? (float)k_min : t;
        t = t > (float)k_max ? (float)k_max : t;
        lvl_out[i] = (int32_t)t - k_min;
    }
}

/* roi_align, aligned=False.  feat: (B, C, H, W); rois: (K, 5) = [batch, x1, y1, x2, y2];
 * out: (K, C, ph, pw).  Rows whose `sel` entry (if sel != NULL) differs from `sel_value`
 * are left untouched (that is how the multi-scale wrapper fills one level at a time). */
void orc_roi_align(const float *feat, int B, int C, int H, int W, const float *rois, int K,
                   int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                   const int32_t *sel, int sel_value, float *out)
{
    (void)B;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 4)
    for (int r = 0; r < K; ++r) {
        if (sel && sel[r] != sel_value) continue;
        const float *roi = rois + 5 * r;
        int b = (int)roi[0];
        float roi_start_w = roi[1] * spatial_scale;
        float roi_start_h = roi[2] * spatial_scale;
        float roi_end_w = roi[3] * spatial_scale;
        float roi_end_h = roi[4] * spatial_scale;
        float roi_width = roi_end_w - roi_start_w;
        float roi_height = roi_end_h - roi_start_h;
        roi_width = roi_width > 1.0f ? roi_width : 1.0f;
        roi_height = roi_height > 1.0f ? roi_height : 1.0f;
        float bin_h = roi_height / (float)pooled_h;
        float bin_w = roi_width / (float)pooled_w;
        int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)pooled_h);
        int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)pooled_w);
        float count = (float)(grid_h * grid_w > 1 ? grid_h * grid_w : 1);
        const float *fb = feat + (size_t)b * C * H * W;
        float *ob = out + (size_t)r * C * pooled_h * pooled_w;
        for (int c = 0; c < C; ++c) {
            const float *f = fb + (size_t)c * H * W;
            for (int ph = 0; ph < pooled_h; ++ph) {
                for (int pw = 0; pw < pooled_w; ++pw) {
                    float acc = 0.0f;
                    for (int iy = 0; iy < grid_h; ++iy) {
                        float yy = roi_start_h + (float)ph * bin_h +
                                   ((float)iy + 0.5f) * bin_h / (float)grid_h;
                        for (int ix = 0; ix < grid_w; ++ix) {
                            float xx = roi_start_w + (float)pw * bin_w +
                                       ((float)ix + 0.5f) * bin_w / (float)grid_w;
                            float y = yy, x = xx;
                            if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
                            if (y <= 0.0f) y = 0.0f;
                            if (x <= 0.0f) x = 0.0f;
                            int y_low = (int)y, x_low = (int)x, y_high, x_high;
                            if (y_low >= H - 1) {
                                y_high = y_low = H - 1;
                                y = (float)y_low;
                            } else {
                                y_high = y_low + 1;
                            }
                            if (x_low >= W - 1) {
                                x_high = x_low = W - 1;
                                x = (float)x_low;
                            } else {
                                x_high = x_low + 1;
                            }
                            float ly = y - (float)y_low, lx = x - (float)x_low;
                            float hy = 1.0f - ly, hx = 1.0f - lx;
                            float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                            acc += w1 * f[y_low * W + x_low] + w2 * f[y_low * W + x_high] +
                                   w3 * f[y_high * W + x_low] + w4 * f[y_high * W + x_high];
                        }
                    }
                    ob[((size_t)c * pooled_h + ph) * pooled_w + pw] = acc / count;
                }
            }
        }
    }
}

/* Backward of roi_align above with respect to `feat`, as torchvision's roi_align_backward kernel defines
 * it (what autograd runs through ROIHead/roi_pooling.py:51-53 when the RoI losses train the backbone):
 * every sample of a bin sends grad * w_k / count to its four taps; samples outside [-1, H] x [-1, W] send
 * nothing; rois get no gradient.  grad_feat (B, C, H, W) is ACCUMULATED into (the caller zeroes it; the
 * multi-level wrapper calls once per level with `sel`).  Accumulation order: RoI by RoI in fp32. */
void orc_roi_align_backward(const float *grad_out, int B, int C, int H, int W, const float *rois, int K,
                            int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                            const int32_t *sel, int sel_value, float *grad_feat)
{
    (void)B;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int c = 0; c < C; ++c) { /* one thread per channel: no two threads add into the same plane */
        for (int r = 0; r < K; ++r) {
            if (sel && sel[r] != sel_value) continue;
            const float *roi = rois + 5 * r;
            int b = (int)roi[0];
            float roi_start_w = roi[1] * spatial_scale;
            float roi_start_h = roi[2] * spatial_scale;
            float roi_width = roi[3] * spatial_scale - roi_start_w;
            float roi_height = roi[4] * spatial_scale - roi_start_h;
            roi_width = roi_width > 1.0f ? roi_width : 1.0f;
            roi_height = roi_height > 1.0f ? roi_height : 1.0f;
            float bin_h = roi_height / (float)pooled_h;
            float bin_w = roi_width / (float)pooled_w;
            int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)pooled_h);
            int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)pooled_w);
            float count = (float)(grid_h * grid_w > 1 ? grid_h * grid_w : 1);
            float *g = grad_feat + ((size_t)b * C + c) * H * W;
            const float *go = grad_out + ((size_t)r * C + c) * pooled_h * pooled_w;
            for (int ph = 0; ph < pooled_h; ++ph)
                for (int pw = 0; pw < pooled_w; ++pw) {
                    float gbin = go[ph * pooled_w + pw];
                    for (int iy = 0; iy < grid_h; ++iy) {
                        float y = roi_start_h + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)grid_h;
                        for (int ix = 0; ix < grid_w; ++ix) {
                            float x = roi_start_w + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)grid_w;
                            float yy = y;
                            if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;
                            if (yy <= 0.0f) yy = 0.0f;
                            if (x <= 0.0f) x = 0.0f;
                            int y_low = (int)yy, x_low = (int)x, y_high, x_high;
                            if (y_low >= H - 1) {
                                y_high = y_low = H - 1;
                                yy = (float)y_low;
                            } else {
                                y_high = y_low + 1;
                            }
                            if (x_low >= W - 1) {
                                x_high = x_low = W - 1;
                                x = (float)x_low;
                            } else {
                                x_high = x_low + 1;
                            }
                            float ly = yy - (float)y_low, lx = x - (float)x_low;
                            float hy = 1.0f - ly, hx = 1.0f - lx;
                            g[y_low * W + x_low] += gbin * (hy * hx) / count;
                            g[y_low * W + x_high] += gbin * (hy * lx) / count;
                            g[y_high * W + x_low] += gbin * (ly * hx) / count;
                            g[y_high * W + x_high] += gbin * (ly * lx) / count;
                        }
                    }
                }
        }
    }
}

/* torchvision.ops.roi_pool semantics (north_star extra, never called by the reference;
 * PARITY UNPINNED).  argmax: flat h*W+w index inside the channel plane, -1 when empty. */
void orc_roi_pool(const float *feat, int B, int C, int H, int W, const float *rois, int K,
                  int pooled_h, int pooled_w, float spatial_scale, float *out, int32_t *argmax)
{
    (void)B;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 4)
    for (int r = 0; r < K; ++r) {
        const float *roi = rois + 5 * r;
        int b = (int)roi[0];
        int rsw = (int)roundf(roi[1] * spatial_scale);
        int rsh = (int)roundf(roi[2] * spatial_scale);
        int rew = (int)roundf(roi[3] * spatial_scale);
        int reh = (int)roundf(roi[4] * spatial_scale);
        int rw = rew - rsw + 1 > 1 ? rew - rsw + 1 : 1;
        int rh = reh - rsh + 1 > 1 ? reh - rsh + 1 : 1;
        float bin_h = (float)rh / (float)pooled_h;
        float bin_w = (float)rw / (float)pooled_w;
        for (int ph = 0; ph < pooled_h; ++ph) {
            for (int pw = 0; pw < pooled_w; ++pw) {
                int hs = (int)floorf((float)ph * bin_h);
                int ws = (int)floorf((float)pw * bin_w);
                int he = (int)ceilf((float)(ph + 1) * bin_h);
                int we = (int)ceilf((float)(pw + 1) * bin_w);
                hs = hs + rsh < 0 ? 0 : (hs + rsh > H ? H : hs + rsh);
                he = he + rsh < 0 ? 0 : (he + rsh > H ? H : he + rsh);
                ws = ws + rsw < 0 ? 0 : (ws + rsw > W ? W : ws + rsw);
                we = we + rsw < 0 ? 0 : (we + rsw > W ? W : we + rsw);
                int empty = (he <= hs) || (we <= ws);
                for (int c = 0; c < C; ++c) {
                    const float *f = feat + ((size_t)b * C + c) * H * W;
                    float mx = empty ? 0.0f : -FLT_MAX;
                    int mi = -1;
                    for (int h = hs; h < he; ++h)
                        for (int w = ws; w < we; ++w)
                            if (f[h * W + w] > mx) {
                                mx = f[h * W + w];
                                mi = h * W + w;
                            }
                    size_t o = (((size_t)r * C + c) * pooled_h + ph) * pooled_w + pw;
                    out[o] = mx;
                    argmax[o] = mi;
                }
            }
        }
    }
}
