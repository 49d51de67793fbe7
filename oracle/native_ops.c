/*
 * oracle/native_ops.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's native operators.  Nothing in the
 * product (neural-motifs_amd/) may import, link or execute this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only
 * as the checker.
 *
 * Compile with:  gcc -O2 -ffp-contract=off -fno-fast-math  (see oracle/Makefile)
 * so that every expression below is evaluated exactly as written: one IEEE
 * fp32 rounding per operation, no fused multiply-add.  The HIP kernels that are
 * required to be bit-exact (NMS, RoIAlign border tests, mask rasteriser) are
 * compiled the same way.
 *
 * Each function cites the reference file:line it restates
 * (paths relative to /root/reference).
 *
 * PINNED: the mask rasteriser and the float64 IoU against the reference's own .pyx files compiled into oracle/_ref/
 * (make -C oracle ref); NMS and RoIAlign (forward + backward) against the reference's own nms_kernel.cu /
 * roi_align_kernel.cu compiled for the CPU (oracle/build_ref_cuda.py) -- bit-exact on the seeded cases of
 * tests/golden/cuda_ref.npz (tests/test_oracle_ref_cuda.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------
 * NMS.  lib/fpn/nms/src/cuda/nms_kernel.cu:23-31 (devIoU), :33-75 (tile mask:
 * box i suppresses box j>i iff IoU(i,j) > thresh), :113-128 (sequential sweep
 * over the score-sorted boxes; kept indices are positions in the SORTED list).
 * ---------------------------------------------------------------------- */
static inline float orc_iou(const float *a, const float *b)
{
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
    float interS = width * height;
    float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
    float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
    return interS / (Sa + Sb - interS);
}

/* keep_out must hold n ints.  Returns the number kept. */
int orc_nms(const float *boxes_sorted, int n, float thresh, int *keep_out)
{
    unsigned char *removed = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int num_keep = 0;
    for (int i = 0; i < n; i++) {
        if (removed[i]) continue;
        keep_out[num_keep++] = i;
        for (int j = i + 1; j < n; j++) {
            if (orc_iou(boxes_sorted + 4 * i, boxes_sorted + 4 * j) > thresh) removed[j] = 1;
        }
    }
    free(removed);
    return num_keep;
}

/* The literal 64x64 bitmask formulation (nms_kernel.cu:33-75 + :107-128), used
 * to validate orc_nms against the reference's own data structure. */
int orc_nms_bitmask(const float *boxes, int n, float thresh, int *keep_out)
{
    const int tpb = 64;
    int col_blocks = n / tpb + ((n % tpb) > 0);
    if (n == 0) return 0;
    uint64_t *mask = (uint64_t *)calloc((size_t)n * col_blocks, sizeof(uint64_t));
    for (int row_start = 0; row_start < col_blocks; row_start++)
        for (int col_start = 0; col_start < col_blocks; col_start++) {
            int row_size = n - row_start * tpb < tpb ? n - row_start * tpb : tpb;
            int col_size = n - col_start * tpb < tpb ? n - col_start * tpb : tpb;
            for (int tx = 0; tx < row_size; tx++) {
                int cur = tpb * row_start + tx;
                uint64_t t = 0;
                int start = (row_start == col_start) ? tx + 1 : 0;
                for (int i = start; i < col_size; i++)
                    if (orc_iou(boxes + 4 * cur, boxes + 4 * (tpb * col_start + i)) > thresh)
                        t |= 1ULL << i;
                mask[(size_t)cur * col_blocks + col_start] = t;
            }
        }
    uint64_t *remv = (uint64_t *)calloc((size_t)col_blocks, sizeof(uint64_t));
    int num_keep = 0;
    for (int i = 0; i < n; i++) {
        int nblock = i / tpb, inblock = i % tpb;
        if (!(remv[nblock] & (1ULL << inblock))) {
            keep_out[num_keep++] = i;
            uint64_t *p = mask + (size_t)i * col_blocks;
            for (int j = nblock; j < col_blocks; j++) remv[j] |= p[j];
        }
    }
    free(mask);
    free(remv);
    return num_keep;
}

/* ------------------------------------------------------------------------
 * RoIAlign (single-sample bilinear crop).
 * Python side lib/fpn/roi_align/functions/roi_align.py:17-32 normalises the
 * rois: height = (H-1)/spatial_scale evaluated in Python double, then the fp32
 * tensor is divided in place by that scalar (scalar is first rounded to fp32).
 * Kernel: lib/fpn/roi_align/src/cuda/roi_align_kernel.cu:15-80 (forward),
 * :103-170 (backward; the reference uses atomicAdd, here a fixed order).
 * features NCHW [B,C,H,W]; rois [N,5] = (im, x1,y1,x2,y2) UN-normalised;
 * out [N,C,ph,pw].
 * ---------------------------------------------------------------------- */
static void orc_normalise_roi(const float *roi, int H, int W, float spatial_scale,
                              float *x1, float *y1, float *x2, float *y2)
{
    /* Python: height = (data_height - 1) / self.spatial_scale  (double) */
    double height_d = (double)(H - 1) / (double)spatial_scale;
    double width_d = (double)(W - 1) / (double)spatial_scale;
    float height = (float)height_d, width = (float)width_d;
    *x1 = roi[1] / width;
    *y1 = roi[2] / height;
    *x2 = roi[3] / width;
    *y2 = roi[4] / height;
}

void orc_roi_align_fwd(const float *feat, int B, int C, int H, int W,
                       const float *rois, int N, int ph, int pw, float spatial_scale,
                       float *out)
{
    for (int n = 0; n < N; n++) {
        const float *roi = rois + 5 * n;
        int b_in = (int)roi[0];
        float x1, y1, x2, y2;
        orc_normalise_roi(roi, H, W, spatial_scale, &x1, &y1, &x2, &y2);
        for (int d = 0; d < C; d++)
            for (int y = 0; y < ph; y++)
                for (int x = 0; x < pw; x++) {
                    float *o = out + (((size_t)n * C + d) * ph + y) * pw + x;
                    if (b_in < 0 || b_in >= B) continue; /* output stays as the caller zero-filled it */
                    const float height_scale = (ph > 1) ? (y2 - y1) * (H - 1) / (ph - 1) : 0;
                    const float width_scale = (pw > 1) ? (x2 - x1) * (W - 1) / (pw - 1) : 0;
                    const float in_y = (ph > 1) ? y1 * (H - 1) + y * height_scale
                                                : (float)(0.5 * (y1 + y2) * (H - 1));
                    if (in_y < 0 || in_y > H - 1) { *o = 0.f; continue; }
                    const float in_x = (pw > 1) ? x1 * (W - 1) + x * width_scale
                                                : (float)(0.5 * (x1 + x2) * (W - 1));
                    if (in_x < 0 || in_x > W - 1) { *o = 0.f; continue; }
                    const int top_y = (int)floorf(in_y), bottom_y = (int)ceilf(in_y);
                    const float y_lerp = in_y - top_y;
                    const int left_x = (int)floorf(in_x), right_x = (int)ceilf(in_x);
                    const float x_lerp = in_x - left_x;
                    const float *plane = feat + ((size_t)b_in * C + d) * H * W;
                    const float tl = plane[top_y * W + left_x], tr = plane[top_y * W + right_x];
                    const float bl = plane[bottom_y * W + left_x], br = plane[bottom_y * W + right_x];
                    const float top = tl + (tr - tl) * x_lerp;
                    const float bottom = bl + (br - bl) * x_lerp;
                    *o = top + (bottom - top) * y_lerp;
                }
    }
}

/* grad_feat must be zero-filled by the caller (roi_align.py:67-68). */
void orc_roi_align_bwd(const float *grad_out, int B, int C, int H, int W,
                       const float *rois, int N, int ph, int pw, float spatial_scale,
                       float *grad_feat)
{
    for (int n = 0; n < N; n++) {
        const float *roi = rois + 5 * n;
        int b_in = (int)roi[0];
        float x1, y1, x2, y2;
        orc_normalise_roi(roi, H, W, spatial_scale, &x1, &y1, &x2, &y2);
        if (b_in < 0 || b_in >= B) continue;
        for (int d = 0; d < C; d++)
            for (int y = 0; y < ph; y++)
                for (int x = 0; x < pw; x++) {
                    const float g = grad_out[(((size_t)n * C + d) * ph + y) * pw + x];
                    const float height_scale = (ph > 1) ? (y2 - y1) * (H - 1) / (ph - 1) : 0;
                    const float width_scale = (pw > 1) ? (x2 - x1) * (W - 1) / (pw - 1) : 0;
                    const float in_y = (ph > 1) ? y1 * (H - 1) + y * height_scale
                                                : (float)(0.5 * (y1 + y2) * (H - 1));
                    if (in_y < 0 || in_y > H - 1) continue;
                    const float in_x = (pw > 1) ? x1 * (W - 1) + x * width_scale
                                                : (float)(0.5 * (x1 + x2) * (W - 1));
                    if (in_x < 0 || in_x > W - 1) continue;
                    const int top_y = (int)floorf(in_y), bottom_y = (int)ceilf(in_y);
                    const float y_lerp = in_y - top_y;
                    const int left_x = (int)floorf(in_x), right_x = (int)ceilf(in_x);
                    const float x_lerp = in_x - left_x;
                    float *plane = grad_feat + ((size_t)b_in * C + d) * H * W;
                    const float dtop = (1 - y_lerp) * g;
                    plane[top_y * W + left_x] += (1 - x_lerp) * dtop;
                    plane[top_y * W + right_x] += x_lerp * dtop;
                    const float dbottom = y_lerp * g;
                    plane[bottom_y * W + left_x] += (1 - x_lerp) * dbottom;
                    plane[bottom_y * W + right_x] += x_lerp * dbottom;
                }
    }
}

/* ------------------------------------------------------------------------
 * Union-box mask rasteriser.  lib/draw_rectangles/draw_rectangles.pyx:24-67
 * (all fp32; `pooling_size` is an unsigned int promoted to float; minmax is a
 * plain clamp with no NaN handling: pyx:24-25 and the generated C).
 * box_pairs [N,8] -> out [N,2,P,P].
 * ---------------------------------------------------------------------- */
static inline float orc_minmax(float x)
{
    float t = (0L > x) ? (float)0L : x;
    return (1L < t) ? (float)1L : t;
}

void orc_draw_union_boxes(const float *box_pairs, int N, unsigned int P, float *out)
{
    for (int n = 0; n < N; n++) {
        const float *bp = box_pairs + 8 * n;
        float x1_union = fminf(bp[0], bp[4]);
        float y1_union = fminf(bp[1], bp[5]);
        float x2_union = fmaxf(bp[2], bp[6]);
        float y2_union = fmaxf(bp[3], bp[7]);
        float w = x2_union - x1_union;
        float h = y2_union - y1_union;
        for (unsigned int i = 0; i < 2; i++) {
            float x1_box = (bp[0 + 4 * i] - x1_union) * P / w;
            float y1_box = (bp[1 + 4 * i] - y1_union) * P / h;
            float x2_box = (bp[2 + 4 * i] - x1_union) * P / w;
            float y2_box = (bp[3 + 4 * i] - y1_union) * P / h;
            for (unsigned int j = 0; j < P; j++) {
                float y_contrib = orc_minmax((j + 1) - y1_box) * orc_minmax(y2_box - j);
                for (unsigned int k = 0; k < P; k++) {
                    float x_contrib = orc_minmax((k + 1) - x1_box) * orc_minmax(x2_box - k);
                    out[(((size_t)n * 2 + i) * P + j) * P + k] = x_contrib * y_contrib;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------
 * Pairwise IoU / intersection ratio in float64.
 * lib/fpn/box_intersections_cpu/bbox.pyx:21-61 and :71-107.
 * boxes [N,4], query [K,4] -> out [N,K] (zero where no overlap).
 * ---------------------------------------------------------------------- */
void orc_bbox_overlaps(const double *boxes, int N, const double *query, int K, double *out)
{
    memset(out, 0, sizeof(double) * (size_t)N * K);
    for (int k = 0; k < K; k++) {
        double box_area = (query[4 * k + 2] - query[4 * k + 0] + 1) * (query[4 * k + 3] - query[4 * k + 1] + 1);
        for (int n = 0; n < N; n++) {
            double iw = fmin(boxes[4 * n + 2], query[4 * k + 2]) - fmax(boxes[4 * n + 0], query[4 * k + 0]) + 1;
            if (iw > 0) {
                double ih = fmin(boxes[4 * n + 3], query[4 * k + 3]) - fmax(boxes[4 * n + 1], query[4 * k + 1]) + 1;
                if (ih > 0) {
                    double ua = (boxes[4 * n + 2] - boxes[4 * n + 0] + 1) * (boxes[4 * n + 3] - boxes[4 * n + 1] + 1)
                                + box_area - iw * ih;
                    out[(size_t)n * K + k] = iw * ih / ua;
                }
            }
        }
    }
}

void orc_bbox_intersections(const double *boxes, int N, const double *query, int K, double *out)
{
    memset(out, 0, sizeof(double) * (size_t)N * K);
    for (int k = 0; k < K; k++) {
        double box_area = (query[4 * k + 2] - query[4 * k + 0] + 1) * (query[4 * k + 3] - query[4 * k + 1] + 1);
        for (int n = 0; n < N; n++) {
            double iw = fmin(boxes[4 * n + 2], query[4 * k + 2]) - fmax(boxes[4 * n + 0], query[4 * k + 0]) + 1;
            if (iw > 0) {
                double ih = fmin(boxes[4 * n + 3], query[4 * k + 3]) - fmax(boxes[4 * n + 1], query[4 * k + 1]) + 1;
                if (ih > 0) out[(size_t)n * K + k] = iw * ih / box_area;
            }
        }
    }
}
