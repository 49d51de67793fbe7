/*
 * motifs_hip.h -- C ABI of libmotifs_hip.so, the MI355X (gfx950) implementation of the
 * neural-motifs per-image hot path.
 *
 * Conventions (SURVEY.md §8b "what the C-ABI replacement must export"):
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless the name ends in _host
 *   - the caller owns every buffer, including outputs and workspaces (sizes from mh_*_ws_bytes)
 *   - `stream` is a hipStream_t passed as void*; nothing here synchronises the device or the stream
 *   - return value: 0 = ok, otherwise a hipError_t (>0) or MH_EINVAL (-1) for bad arguments;
 *     nothing calls exit()
 *   - the device is whatever the caller made current (hipSetDevice); entry points are reentrant
 *
 * Each entry point cites the reference interface it replaces (paths under the reference repo).
 */
#ifndef MOTIFS_HIP_H
#define MOTIFS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_OK 0
#define MH_EINVAL (-1)
#define MH_EUNSUPPORTED (-2)   /* entry point not available in this build (caller uses the documented alternative) */
#define MH_EFAULT (-3)         /* an earlier persistent-kernel launch on this device reported a device-side fault */

/* epilogue flags for GEMM / conv */
#define MH_EPI_NONE 0
#define MH_EPI_RELU 1
#define MH_EPI_RELU6 2
/* OR-ed into `epilogue` of mh_plconv3x3 / _to_image / _pool_to_image (round 6): the first 64 KiB of `workspace` (the arrival
 * counters of the sliced tiles) were zero when the buffer was first handed to the library and have only been used by these calls
 * since (every launch leaves them zero): the call skips its memset. */
#define MH_EPI_WS_ZEROED 0x100

int mh_version(void);
/* how the fp32 matrix products of mh_gemm_* / mh_conv3x3_* are evaluated (MFMAs per fp32 product): always 3, with
 * mh_split_f16() == 1 = f16x3: each operand row is scaled by a power of two (its largest magnitude lands in
 * [2^14, 2^15)), every element splits into two f16 terms h1 + h2 (|remainder| <= 2^-24 |a|), three f16 MFMAs
 * (h1*h1 + h1*h2 + h2*h1) accumulate in fp32, the scales come off exactly in the epilogue.  (Rounds 1-2 also shipped a
 * bf16x6 and an f32-input-MFMA build; their code is gone, their measured error against float64 is kept in
 * profiles/r02_split_check.jsonl, tools/split_check.cpp.) */
int mh_mfma_split(void);
/* always 0 (kept for ABI stability: the bf16x6 builds it described were removed in round 3) */
int mh_split_rne(void);
/* 1 in the f16x3 build (the default); workspaces of mh_gemm_f32 / mh_conv3x3_* then also hold the row exponents */
int mh_split_f16(void);
/* name of the last kernel-launch error on this thread (for diagnostics), or "" */
const char *mh_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * NMS.  Replaces `int nms_apply(THIntTensor* keep, THCudaTensor* boxes_sorted, float thresh)`
 * (lib/fpn/nms/src/nms_cuda.h:1, nms_kernel.cu:88-132).  Fully on device: a 64x64-tile wavefront
 * bitmask IoU kernel followed by a single-wave sweep; no D2H copy, no hipMalloc.
 *   boxes_sorted [n,4] fp32 xyxy, sorted by descending score
 *   keep         [n]   int32 out: kept positions (into the sorted list), ascending
 *   num_keep     [1]   int32 out
 *   workspace    >= mh_nms_ws_bytes(n)
 * mh_nms_batched runs `nseg` independent problems (one per (image,class) segment): segment s covers
 * boxes [seg_offsets[s], seg_offsets[s+1]); keep/num_keep are per segment (keep uses the same
 * offsets, positions are segment-relative).  max_seg = largest segment length (host value).
 * ------------------------------------------------------------------------------------------- */
size_t mh_nms_ws_bytes(int n);
int mh_nms(const float *boxes_sorted, int n, float thresh, int *keep, int *num_keep,
           void *workspace, size_t ws_bytes, void *stream);
size_t mh_nms_batched_ws_bytes(int total_boxes, int nseg, int max_seg);
int mh_nms_batched(const float *boxes_sorted, const int *seg_offsets, int nseg, int total_boxes,
                   int max_seg, float thresh, int *keep, int *num_keep, void *workspace,
                   size_t ws_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * RoIAlign (single-sample bilinear crop).  Replaces roi_align_forward_cuda / roi_align_backward_cuda
 * (lib/fpn/roi_align/src/roi_align_cuda.h:1-6) INCLUDING the roi normalisation the Python wrapper
 * did (functions/roi_align.py:17-32): rois are passed un-normalised, [n,5] = (im, x1,y1,x2,y2).
 *   feat_layout 0: feat is NCHW [B,C,H,W] (the reference layout)
 *   feat_layout 1: feat is NHWC [B,H,W,C] (the trunk's internal layout; coalesced over C)
 *   out [n,C,ph,pw] always (the layout fc6 expects).  Rows whose image index is out of range are
 *   zero-filled.  The backward OVERWRITES grad_feat (same layout flag); like the reference it scatters
 *   with fp32 atomicAdd, so its summation order is not reproducible run to run (it is not on the
 *   train_rels path: the feature map is detached there, lib/rel_model.py:491).
 * ------------------------------------------------------------------------------------------- */
int mh_roi_align_fwd(const float *feat, int B, int C, int H, int W, int feat_layout,
                     const float *rois, int n, int ph, int pw, float spatial_scale, float *out,
                     void *stream);
int mh_roi_align_bwd(const float *grad_out, int B, int C, int H, int W, int feat_layout,
                     const float *rois, int n, int ph, int pw, float spatial_scale,
                     float *grad_feat, void *stream);
/* Deterministic variant of the backward (same arguments): a gather per input pixel in a fixed (roi, bin, corner) order
 * instead of the reference's atomicAdd scatter (roi_align_kernel.cu:157-168) -- bit-reproducible run to run; used by the
 * detector pre-training path (models/train_detector.py), where the gradient flows into the trunk.  C <= 1024. */
int mh_roi_align_bwd_det(const float *grad_out, int B, int C, int H, int W, int feat_layout, const float *rois, int n,
                         int ph, int pw, float spatial_scale, float *grad_feat, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Union-box mask rasteriser.  Replaces the Cython `draw_union_boxes(bbox_pairs[N,8], pooling_size)`
 * (lib/draw_rectangles/draw_rectangles.pyx:12-67) and the host round trip around it
 * (lib/get_union_boxes.py:47-50).  out = mask + offset (the caller passes -0.5f, get_union_boxes.py:49).
 *   channels_last 0: out [n,2,P,P]   1: out [n,P,P,2]
 * ------------------------------------------------------------------------------------------- */
int mh_draw_union_boxes(const float *box_pairs, int n, int P, float offset, int channels_last,
                        float *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Relation tail (round 6): prod[r] = subj[i1[r]] * obj[i2[r]] (* vis[r]) -- replaces the two row gathers and the two
 * multiplies of /root/reference lib/rel_model.py:500-512 (`subj_rep[rel_inds[:, 1]] * obj_rep[rel_inds[:, 2]]`, then `* vr`)
 * and, in the backward pass, their four multiplies, two sort-based index-add chains and two select-backward fills.
 *   edge [n][2][D] fp32: subject (side 0) and object (side 1) representation of every box; i1 / i2 [R] int64; vis [R][D] or NULL
 *   forward : out [R][D]
 *   backward: d_vis [R][D] = grad_out * (subj * obj)   (NULL iff vis is NULL);
 *             d_edge [n][2][D]: side 0 of box i = sum over the rows r with i1[r] == i of (grad_out[r] * vis[r]) * obj[i2[r]],
 *             side 1 over the rows with i2[r] == i of (grad_out[r] * vis[r]) * subj[i1[r]].  The rows of a box are given by the
 *             caller: order [2][R] (side-major row lists) and ptr [2][n+1] (offsets INTO order, both sides in one index space:
 *             ptr[1][*] >= R); sums run in list order -- deterministic, no atomics.  D % 4 == 0, pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
int mh_pair_product_fwd(const float *edge, int n, int D, const long long *i1, const long long *i2, int R, const float *vis,
                        float *out, void *stream);
int mh_pair_product_bwd(const float *edge, int n, int D, const long long *i1, const long long *i2, int R, const float *vis,
                        const float *grad_out, const int *order, const int *ptr, float *d_edge, float *d_vis, void *stream);

/* Relation tail (round 6): out[r] = logits[r] + table[labels[i1[r]] * num_objs + labels[i2[r]]] -- the FrequencyBias term of
 * /root/reference lib/rel_model.py:528-531 (`self.freq_bias.index_with_labels(torch.stack((obj_preds[rel_inds[:, 1]],
 * obj_preds[rel_inds[:, 2]]), 1))`, lib/sparse_targets.py:39-45) in one launch; keys [R] (int64, out) = the table rows used.
 * mh_freq_bias_bwd: d_table [table_rows][P] = sum of grad_out rows per key, in ascending row order (deterministic: the first row of
 * a key gathers the rows of its key; no device sort, no atomics); d_table is cleared by the call.  R <= 8192. */
int mh_freq_bias_add(const float *logits, const float *table, const long long *labels, const long long *i1, const long long *i2, int R,
                     int P, int num_objs, float *out, long long *keys, void *stream);
int mh_freq_bias_bwd(const float *grad_out, const long long *keys, int R, int P, long long table_rows, float *d_table, void *stream);

/* The training script's two cross-entropy losses (round 6; /root/reference models/train_rels.py:140-141:
 * `F.cross_entropy(result.rm_obj_dists, result.rm_obj_labels)`, `F.cross_entropy(result.rel_dists, result.rel_labels[:, -1])`) in
 * two launches, their backward in one.  logits_x [Rx][Cx] fp32 contiguous; label of row r at labels_x[r * stride_x] (int64,
 * 0 <= label < Cx).  Forward: lse [Ra + Rb] (log-sum-exp per row, kept for the backward pass), rowloss [Ra + Rb] (scratch),
 * losses [2] = the two means (summed in a fixed order).  Backward: grad_x[r][c] = (exp(x - lse_r) - [c == label_r]) * upstream[x] / Rx,
 * upstream [2] on the device; a NULL grad pointer skips that side.  A label outside [0, Cx) is the caller's error: nothing is read
 * outside the row and that side's loss is NaN (the framework's kernel asserts on the device; ignore_index is not implemented -- the
 * reference's labels never use it). */
int mh_ce_pair_fwd(const float *logits_a, const long long *labels_a, long long stride_a, int Ra, int Ca, const float *logits_b,
                   const long long *labels_b, long long stride_b, int Rb, int Cb, float *lse, float *rowloss, float *losses, void *stream);
int mh_ce_pair_bwd(const float *logits_a, const long long *labels_a, long long stride_a, int Ra, int Ca, const float *logits_b,
                   const long long *labels_b, long long stride_b, int Rb, int Cb, const float *lse, const float *upstream, float *grad_a,
                   float *grad_b, void *stream);

/* fp32 pairwise IoU, torch semantics of lib/fpn/box_utils.py:85-131: out[a,b] */
int mh_bbox_overlaps(const float *boxes_a, int na, const float *boxes_b, int nb, float *out,
                     void *stream);

/* Recall@K triplet matching on the device (the step after the path: lib/evaluation/sg_eval.py:243-284,
 * _compute_pred_matches).  gt_triplets / pred_triplets [n,3] = (subject class, predicate, object class);
 * gt_boxes / pred_boxes [n,8] = subject box, object box; float64 IoU with the reference's +1 convention.
 * first_match [G]: smallest matching prediction index (INT_MAX: none); nmatch [P]: matches per prediction. */
int mh_triplet_match(const int *gt_triplets, const float *gt_boxes, int G, const int *pred_triplets,
                     const float *pred_boxes, int P, double iou_thresh, int *first_match, int *nmatch,
                     void *stream);

/* ---------------------------------------------------------------------------------------------
 * FP32 GEMM on MFMA (f16x3 on v_mfma_f32_32x32x16_f16; see mh_mfma_split).  Replaces the cuBLAS / nn.Linear
 * calls on the path (lib/object_detector.py:80-104, lib/rel_model.py:367-390,
 * highway_lstm_kernel.cu:441-465).  Row-major:
 *     C[M,N] = epi( opA(A)[M,K] * opB(B)[K,N] + bias[N] )   (+ C if accumulate)
 *   transA 0: A stored [M,K] (lda)   1: A stored [K,M]
 *   transB 0: B stored [K,N] (ldb)   1: B stored [N,K]     (nn.Linear weight -> transB=1)
 *   bias may be NULL.  splitk <= 0 lets the library choose.  The workspace is MANDATORY (>= mh_gemm_ws_bytes, 256-byte
 *   aligned): since round 3 the call first writes both operands as f16 plane images into it (below), then multiplies those.
 * ------------------------------------------------------------------------------------------- */
size_t mh_gemm_ws_bytes(int M, int N, int K, int splitk);
int mh_gemm_auto_splitk(int M, int N, int K);
int mh_gemm_f32(int transA, int transB, int M, int N, int K, const float *A, int lda,
                const float *B, int ldb, float *C, int ldc, const float *bias, int epilogue,
                int accumulate, int splitk, void *workspace, size_t ws_bytes, void *stream);

/* Plane images (round 3; csrc/pl_tile.h): an operand with `rows` rows (the non-K index) and K columns stored as
 * cell(kc, r) = 64 bytes at (kc * rows + r) * 64 = h1[16] | h2[16] (f16 terms of x * 2^e_r for k = 16 kc ..), followed
 * (256-byte aligned) by the rows' largest |x| as fp32 bit patterns.  An image is made ONCE per operand value and reused by
 * every product that reads the operand in that orientation (cached weight images: lib/hip_ops.py) -- the K loop of
 * mh_gemm_planes then contains no split arithmetic at all.
 *   mh_make_planes: k_contiguous 1: X is [rows][K] (ld); 0: X is [K][rows] (ld) -- the transposition happens here, once.
 *   mh_gemm_planes: C[M,N] = epi(A[M,K] . B[N,K]^T + bias) (+ C); images 256-byte aligned, < 2 GiB each.
 * Replaces the same cuBLAS calls as mh_gemm_f32 (lib/rel_model.py:366-373,403-414, lib/object_detector.py:129-138). */
size_t mh_planes_bytes(long long rows, long long K);
int mh_make_planes(const float *X, int k_contiguous, long long rows, long long K, long long ld, void *image, void *stream);
/* both images of X [R][C] (ld) from one HBM read: img_rows (rows = X rows, K = C; mh_planes_bytes(R, C)) and img_cols
 * (rows = X columns, K = R; mh_planes_bytes(C, R)) -- a Linear layer's weight / input / output gradient are each consumed in
 * both orientations (forward + input gradient, forward + weight gradient, input + weight gradient). */
int mh_make_planes_both(const float *X, long long R, long long C, long long ld, void *img_rows, void *img_cols, void *stream);
size_t mh_gemm_planes_ws_bytes(int M, int N, int K, int splitk);
int mh_gemm_planes_auto_splitk(int M, int N, int K);
int mh_gemm_planes(int M, int N, int K, const void *A_image, const void *B_image, float *C, int ldc, const float *bias,
                   int epilogue, int accumulate, int splitk, void *workspace, size_t ws_bytes, void *stream);
/* A/B hooks for measurements (tools/pl_check.cpp): force a block-tile shape (-1 auto; round-3 loop: 0 256x128, 1 128x128, 2 256x64;
 * ring loop: 3 256x256 on eight waves, 4 256x128);
 * the round-2 in-loop-split kernel (fp32 operands split inside the K loop) kept for comparison runs. */
void mh_debug_pl_shape(int shape);
/* tile order of the plane GEMMs: 1 (default) = XCD-banded (K slices on their own XCDs, bands of the tile space otherwise:
 * csrc/pl_gemm.hip gemm_item), 0 = the round-3 patch numbering (A/B and traffic runs; MH_GEMM_ORDER=0 does the same per process) */
void mh_debug_pl_order(int order);
/* host-side replay of the kernels' block -> (tile, K slice) mapping for the launch mh_gemm_planes(M, N, K, splitk) would make:
 * out[10] = {grid_x, grid_y, tm, tn, z, valid, tiles_m, tiles_n, splitk, order} (no device needed; tests/test_gemm_order.py) */
int mh_debug_pl_item(int M, int N, int K, int splitk, long long block, int *out);
size_t mh_gemm_ws_bytes_v2(int M, int N, int K, int splitk);
int mh_gemm_auto_splitk_v2(int M, int N, int K);
int mh_gemm_f32_v2(int transA, int transB, int M, int N, int K, const float *A, int lda,
                   const float *B, int ldb, float *C, int ldc, const float *bias, int epilogue,
                   int accumulate, int splitk, void *workspace, size_t ws_bytes, void *stream);
/* The small-product engine (round 5; csrc/gemm.hip): ONE launch per product for everything mh_gemm_f32 does not put on plane
 * images -- the ~40 small / skinny nn.Linear-shaped products of a step (lib/rel_model.py:171-296,500-524: obj_embed, pos_embed,
 * post_lstm, rel_compress, the trainable object fc6/fc7; lib/lstm/decoder_rnn.py:96-131; their input and weight gradients).
 * Same contract as mh_gemm_f32 (fp32 operands in any of the four orientations, bias / activation / accumulate), evaluated as
 * "bf16x6": each fp32 operand element is the exact sum of three bf16 terms (round to nearest), six v_mfma_f32_32x32x16_bf16
 * per accumulator and k-tile, fp32 accumulate -- no power-of-two row scales, hence no pass over the operands for their row
 * maxima in front of the product (what f16x3 needs).  mh_gemm_f32_v2 = the same engine without arrival counters.
 *   counters / n_counters   split-K arrival counters, one int per 128x128 (N <= 64: 256x64) output tile: ZERO on entry, left
 *                           ZERO on return; the last K slice of a tile to finish adds the slices in slice order (bit-identical
 *                           for any arrival order) and applies the epilogue inside the GEMM launch.  One array per HIP stream
 *                           (products on different streams may run concurrently).  NULL / too few: the reduction is a second
 *                           launch.  mh_gemm_small_max_counters() = the largest number ever used.
 *   workspace               >= mh_gemm_ws_bytes_v2(M, N, K, splitk) (split-K partial sums), 256-byte aligned */
int mh_gemm_small_max_counters(void);
int mh_debug_small_plan(int M, int N, int K);   /* tests / tools: K slices | (1 << 16 if the reduction is fused) for an auto split */
int mh_gemm_small_f32(int transA, int transB, int M, int N, int K, const float *A, int lda,
                      const float *B, int ldb, float *C, int ldc, const float *bias, int epilogue,
                      int accumulate, int splitk, void *workspace, size_t ws_bytes, int *counters,
                      int n_counters, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The trunk's 3x3 convolutions on the plane engine (round 3; csrc/pl_conv.hip).  Replaces cuDNN for
 * vgg16.features (lib/object_detector.py:110-118, :623-633) on the frozen-trunk path of models/train_rels.py.
 *   activation image  cells [C/16][B*H*W][64 B] (h1 | h2 of x * 2^e_b, e_b from the largest |x| of image b), followed
 *                     (256-byte aligned) by maxbits[B]; written by mh_act_planes from an fp32 NHWC tensor and the
 *                     per-image maxima its producer reported, optionally through the 2x2/2 max-pool (pool = 1).
 *   packed weights    cells [tap][Cin/16][Cout][64 B] + maxbits[Cout] (mh_plconv_pack_weight; flip_transpose = the
 *                     input-gradient conv's weights, as for mh_conv3x3_pack_weight).
 *   mh_plconv3x3      out [B,H,W,Cout] fp32 = epi(conv3x3(image, packed) + bias); out_maxbits[B] (may be NULL; must be
 *                     ZERO on entry) receives the largest |out| per image.  Workspace: mh_plconv3x3_ws_bytes (split-K
 *                     partial sums of the tile schedule).
 *   mh_conv_first_nchw_max = mh_conv_first_nchw that also reports those maxima (conv1_1 feeds the first image).
 * mh_debug_plconv_shape: A/B hook (-1 auto; round-3 loop: 0 256x128, 1 128x128, 2 256x64 block tiles; 4: the ring kernel's
 * 256x128, what auto selects for Cout >= 128). */
size_t mh_act_planes_bytes(int B, int H, int W, int C);
/* per-image maxima (fp32 bit patterns) of a [B][n] fp32 tensor, for mh_act_planes of a tensor whose producer did not report them.
 * Round 5: mh_act_planes and mh_plconv3x3 (fp32 output) take up to 65535 images -- the 1536 7x7 RoI maps of the mask tower's 3x3
 * conv (lib/get_union_boxes.py:35-37) and of the ResNet layer4 stacks (lib/resnet.py:126-133) run on the ring engine that way. */
int mh_image_maxbits(const float *x, int B, long long n_per_image, unsigned *bits, void *stream);
int mh_act_planes(const float *x_nhwc, const unsigned *maxbits, int B, int H, int W, int C, int pool, void *image,
                  void *stream);
size_t mh_plconv_packed_bytes(int Cout, int Cin);
int mh_plconv_pack_weight(const float *w, int Cout, int Cin, int flip_transpose, void *packed, void *stream);
size_t mh_plconv3x3_ws_bytes(int B, int H, int W, int Cin, int Cout);
int mh_plconv3x3(const void *in_image, int B, int H, int W, int Cin, const void *packed, int Cout, const float *bias,
                 int epilogue, float *out, unsigned *out_maxbits, void *workspace, size_t ws_bytes, void *stream);
int mh_conv_first_nchw_max(const float *in_nchw, int B, int Cin, int H, int W, const float *w, int Cout,
                           const float *bias, int epilogue, float *out_nhwc, unsigned *maxbits, void *stream);
/* The same convolutions with an IMAGE output: the epilogue splits what it computed and writes the next layer's activation
 * image itself (mh_act_planes_bytes(B, H, W, Cout)) -- no fp32 tensor, no converter pass.  The image's per-image scale comes
 * from the bound |y| <= max|x_b| * max_n sum|w_n| + max|bias| (the true maximum is not known before the data is written);
 * in_true_maxbits [B] = the TRUE per-image maxima of the input as reported by its producer, out_maxbits [B] (zero on entry)
 * receives the true maxima of the output for the next layer's bound.
 *   mh_stem_to_image: conv1_1 (NCHW image, Cin <= 4, B <= 32) straight to the image conv1_2 reads. */
int mh_plconv3x3_to_image(const void *in_image, const unsigned *in_true_maxbits, int B, int H, int W, int Cin,
                          const void *packed, int Cout, const float *bias, int epilogue, void *out_image,
                          unsigned *out_maxbits, void *workspace, size_t ws_bytes, void *stream);
/* the same conv through the 2x2 / 2 max-pool that follows it (H, W even): out_image = activation image of the POOLED output
 * [B, H/2, W/2, Cout], mh_act_planes_bytes(B, H / 2, W / 2, Cout) bytes; the pool happens in the kernel's epilogue (tile rows in
 * pool order), bit-identical to mh_plconv3x3 followed by mh_act_planes(pool = 1).  Replaces the nn.MaxPool2d modules of
 * vgg16.features behind conv1_2 / conv2_2 / conv3_3 / conv4_3 (reference lib/object_detector.py:110-118, :623-626). */
int mh_plconv3x3_pool_to_image(const void *in_image, const unsigned *in_true_maxbits, int B, int H, int W, int Cin, const void *packed,
                               int Cout, const float *bias, int epilogue, void *out_image, unsigned *out_maxbits, void *workspace,
                               size_t ws_bytes, void *stream);
int mh_stem_to_image(const float *in_nchw, int B, int Cin, int H, int W, const float *w, int Cout, const float *bias,
                     int epilogue, void *out_image, unsigned *out_maxbits, void *stream);

void mh_debug_plconv_shape(int shape);
void mh_debug_plconv_splitk(int splitk);   /* 0 = the planner's schedule; > 0 = every tile in that many K slices (sweeps) */
void mh_debug_plconv_flags(int flags);     /* measurement only; bit 1: the ring kernel returns without its epilogue (no output) */

/* ---------------------------------------------------------------------------------------------
 * Convolution stack, NHWC internal layout (cuDNN replacement; lib/object_detector.py:110-118,
 * :503-508, lib/get_union_boxes.py:31-39).
 *   mh_conv3x3_nhwc: 3x3, stride 1, pad 1 implicit GEMM on MFMA, fused bias + ReLU/ReLU6.
 *       in [B,H,W,Cin] (Cin % 16 == 0), wt = packed weights of THIS conv (see mh_conv3x3_pack_weight),
 *       out [B,H,W,Cout] (Cout % 4 == 0)
 *   mh_conv3x3_pack_weight: w [Cout,Cin,3,3] (API layout) -> wt, mh_conv3x3_packed_floats(N, K) floats for a conv
 *       with N output and K input channels: per (tap, output channel, 16 input channels) the 16-bit planes of the split
 *       -- h1|h2 (f16, 64 B) scaled by the output channel's power of two, whose exponents follow the planes, in the
 *       default f16x3 build; hi|mid|lo (bf16, 96 B) in the bf16x6 build; plain fp32 [9][N][K] in the f32-MFMA build.
 *       Always size the buffer with mh_conv3x3_packed_floats.  flip_transpose=1 produces the weights of the dgrad conv (N = Cin, K = Cout, taps mirrored)
 *   mh_conv_first_nchw: the 3->Cout stem reading the NCHW image directly, writing NHWC; bias+ReLU
 *   mh_maxpool2x2_nhwc: 2x2/2 max pool (floor), NHWC
 *   mh_im2col_nhwc: generic patch matrix out[B*Ho*Wo, ldo] with column (ky*kw+kx)*C + c
 *   mh_col2im... not needed on this path (the only im2col conv takes an input without gradient)
 *   mh_nchw_to_nhwc / mh_nhwc_to_nchw: layout converters
 * ------------------------------------------------------------------------------------------- */
size_t mh_conv3x3_packed_floats(int Cout, int Cin);
int mh_conv3x3_pack_weight(const float *w, int Cout, int Cin, int flip_transpose, float *wt,
                           void *stream);
size_t mh_conv3x3_ws_bytes(int B, int H, int W, int Cin, int Cout);   /* partial-sum scratch (0 if no tile is split) */
/* The tile schedule mh_conv3x3_nhwc uses for this shape (diagnostics / tests; pure host arithmetic):
 * out8_host = {block rows BM, block cols BN, m-tiles, n-tiles, uniform K split of the body tiles, body m-tiles,
 * tail tiles, K slices per tail tile}.  Tiles beyond the last full round of resident blocks (2 per CU) form the tail
 * and are cut along K, so that the leftover occupies the whole chip briefly instead of a few CUs for a block time. */
int mh_conv3x3_schedule(int B, int H, int W, int Cin, int Cout, int *out8_host);
int mh_conv3x3_nhwc(const float *in, int B, int H, int W, int Cin, const float *wt, int Cout,
                    const float *bias, int epilogue, float *out, void *workspace, size_t ws_bytes,
                    void *stream);
/* replaces cuDNN's convolution weight gradient reached through nn.Conv2d autograd (mask tower conv,
 * lib/get_union_boxes.py:31-39; every VGG / RPN conv when the detector trains, models/train_detector.py:141-146):
 * weight gradient of the 3x3/1/1 conv as an implicit GEMM over the pixels (no patch matrix): dw [Cout][9*Cin]
 * (tap-major, then input channel) from x [B,H,W,Cin] and gy [B,H,W,Cout]; Cin, Cout % 4 == 0.  MH_EUNSUPPORTED in the
 * f32-MFMA build (use mh_im2col_nhwc + mh_gemm_f32 there). */
size_t mh_conv3x3_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout);
int mh_conv3x3_wgrad(const float *x, const float *gy, int B, int H, int W, int Cin, int Cout, float *dw,
                     void *workspace, size_t ws_bytes, void *stream);
int mh_conv_first_nchw(const float *in_nchw, int B, int Cin, int H, int W, const float *w /*[Cout,Cin,3,3]*/,
                       int Cout, const float *bias, int epilogue, float *out_nhwc, void *stream);
int mh_maxpool2x2_nhwc(const float *in, int B, int H, int W, int C, float *out, void *stream);
/* detector pre-training (models/train_detector.py; the trunk is trainable there; replaces the autograd backward of
 * nn.MaxPool2d / nn.ReLU in torchvision's vgg16.features, lib/object_detector.py:623-633):
 *   mh_maxpool2x2_bwd_nhwc: gradient to the first maximal element of each window (torch semantics), gin fully written
 *   mh_act_bwd: gradient through a fused ReLU / ReLU6 epilogue given the activated output y */
int mh_maxpool2x2_bwd_nhwc(const float *in, const float *gout, int B, int H, int W, int C, float *gin,
                           void *stream);
int mh_act_bwd(const float *g, const float *y, long long n, int epilogue, float *out, void *stream);
int mh_im2col_nhwc(const float *in, int B, int H, int W, int C, int kh, int kw, int stride, int pad,
                   float *out, int ldo, void *stream);
int mh_nchw_to_nhwc(const float *in, int B, int C, int H, int W, float *out, void *stream);
int mh_nhwc_to_nchw(const float *in, int B, int C, int H, int W, float *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Stacked alternating-direction highway LSTM.  Replaces highway_lstm_forward_cuda /
 * highway_lstm_backward_cuda (lib/lstm/highway_lstm_cuda/src/highway_lstm_cuda.h:1-14) with the same
 * caller-owned buffer set, minus tmp_i/tmp_h (superseded by the workspace) and plus an explicit stream.
 *   x        [T,B,in]  zero padded, sequences sorted by decreasing length
 *   lengths_host [B]   int32, HOST memory (as in the reference)
 *   h_data, c_data [L,T+1,B,H]  zero-filled by the caller (slot 0 = initial state)
 *   weight   flat: per layer Wx[in_l,6H] then Wh[H,5H];  bias flat [L,5H]
 *   dropout  [L,B,H]  (scaled keep mask, shared over time)
 *   gates    [L,T,B,6H] written when is_training (may be NULL otherwise)
 * The input projection x_t*Wx is hoisted out of the time loop into one MFMA GEMM per layer; the
 * recurrence of a whole layer is ONE persistent launch (a block owns 4 hidden units for all timesteps, Wh slice in
 * registers; the state travels between the workgroups as tagged 8-byte granules swept straight into LDS -- no grid
 * barrier; MH_LSTM_GRAN=0 selects the agent-scope barrier per step of rounds 2-3) when H <= 512, H % 4 == 0, B <= 32;
 * other shapes run one fused GEMV+gate kernel per (layer,t).
 * Fault behaviour of the persistent launches (the reference only fprintf's CUDA errors, highway_lstm_kernel.cu:17-29):
 * every wait is bounded (the granule sweep by 400 ms of the device's wall clock, the barrier spin by its count); a block
 * whose wait expires stores 1 into a host-pinned fault word, the launch poisons its
 * outputs (h / gate gradients) with NaN, and EVERY later mh_hwlstm_* / mh_hwcell_seq_* call on that device returns
 * MH_EFAULT until mh_fault_clear().  mh_fault_pending() is a host read (no synchronisation): poll it at step end.
 * Backward: out_grad [T,B,H]; outputs x_grad [T,B,in] (overwritten); when do_weight_grad: weight_grad (OVERWRITTEN since
 * round 5: every region of the flat vector is written exactly once per call -- no 64 MB zero fill in front) and bias_grad
 * (ACCUMULATED into, the caller zero-fills its 5*H*L floats).
 * ------------------------------------------------------------------------------------------- */
size_t mh_hwlstm_fwd_ws_bytes(int in_size, int H, int B, int L, int T);
int mh_hwlstm_fwd(int in_size, int H, int B, int L, int T, const float *x,
                  const int *lengths_host, float *h_data, float *c_data, const float *weight,
                  const float *bias, const float *dropout, float *gates, int is_training,
                  void *workspace, size_t ws_bytes, void *stream);
size_t mh_hwlstm_bwd_ws_bytes(int in_size, int H, int B, int L, int T);
int mh_hwlstm_bwd(int in_size, int H, int B, int L, int T, const float *out_grad,
                  const int *lengths_host, const float *x, const float *h_data,
                  const float *c_data, const float *weight, const float *gates,
                  const float *dropout, float *x_grad, float *weight_grad, float *bias_grad,
                  int do_weight_grad, void *workspace, size_t ws_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * One fused highway-LSTM cell step for the label decoder (lib/lstm/decoder_rnn.py:96-131):
 *   pre_i [n,6H] = input projection INCLUDING its bias (hoisted GEMM + embedding-table gather)
 *   h_prev, c_prev [n,H];  wh_t [5H,H] = state_linearity.weight (nn.Linear layout, K contiguous)
 *   bias_h [5H];  dropout [n,H] or NULL;  outputs h_out, c_out [n,H]; gates_out [n,6H] or NULL
 * and its backward (d_h, d_c_out -> gate grads [n,6H], d_c_in [n,H]); the recurrent dgrad/wgrad
 * GEMMs are issued by the caller through mh_gemm_f32 / mh_gemv_rows.
 * ------------------------------------------------------------------------------------------- */
int mh_hwlstm_cell_fwd(int n, int H, const float *pre_i, const float *h_prev, const float *c_prev,
                       const float *wh_t, const float *bias_h, const float *dropout,
                       float *h_out, float *c_out, float *gates_out, void *stream);
int mh_hwlstm_cell_bwd(int n, int H, const float *d_h, const float *d_c_out, const float *c_prev,
                       const float *c_out, const float *gates, const float *dropout,
                       float *d_gates /*[n,6H]*/, float *d_c_in, void *stream);
/* out[n,R] = v[n,K] * Wt[R,K]^T (+bias[R]); n <= 64 small-batch GEMV, one wave per output row */
int mh_gemv_rows(int n, int R, int K, const float *v, int ldv, const float *wt, int ldw,
                 const float *bias, float *out, int ldo, void *stream);

/* One highway-LSTM layer over a PACKED time-major batch (PackedSequence order) in a single launch: the label
 * decoder's recurrence under teacher forcing (lib/lstm/decoder_rnn.py:151-215; the reference steps a Python loop of
 * cuBLAS + elementwise calls).  batch_sizes_host[T] non-increasing, N = sum.  h_buf / c_buf: B + N rows, the first
 * B = the zero initial state (zeroed by the caller), row B + r = state of packed row r.  pre_i [N,6H] = input
 * projection incl. bias; w_state [5H,H]; gates [N,6H] saved for the backward.  Shapes: H <= 512, H % 4 == 0,
 * B <= 32 (MH_EINVAL otherwise -- step with mh_hwlstm_cell_fwd/bwd instead).  Workspace: mh_hwcell_seq_ws_bytes(). */
size_t mh_hwcell_seq_ws_bytes(void);
int mh_hwcell_seq_fwd(int H, int B, int T, const int *batch_sizes_host, const float *pre_i,
                      const float *w_state, const float *b_state, const float *dropout /*[B,H] or NULL*/,
                      float *h_buf, float *c_buf, float *gates, void *workspace, size_t ws_bytes, void *stream);
int mh_hwcell_seq_bwd(int H, int B, int T, const int *batch_sizes_host, const float *dh_all /*[N,H]*/,
                      const float *c_buf, const float *gates, const float *dropout,
                      const float *w_state_t /*[H,5H]*/, float *d_pre /*[N,6H]*/, float *hgrad_buf,
                      float *cgrad_buf /*scratch, B + N rows each*/, void *workspace, size_t ws_bytes,
                      void *stream);

/* The label decoder's greedy pass in ONE persistent launch (lib/lstm/decoder_rnn.py:205-227: the reference steps a Python
 * loop of 3 small GEMMs + arg-max + embedding lookup per object): per step the highway-LSTM cell on
 * enc_proj[row] + emb_proj[label fed], the class logits out = w_out * h + b_out, the arg-max over the non-background
 * classes (ties to the lower class) and the gather of the next step's embedding row, separated by two grid barriers.
 *   batch_sizes_host[T] non-increasing (PackedSequence), N = sum;  enc_proj [N,6H] = input projection of the encoder
 *   part incl. bias;  emb_proj [C+1,6H] = input projection of every label embedding (row 0 = 'start', row l+1 = label l)
 *   labels [N] int64 or NULL: teacher forcing -- a row whose label is non-zero commits that label, a background row its
 *   arg-max (training, :205-213); NULL = pure greedy decoding (evaluation)
 *   outputs: h_buf / c_buf [B+N,H] (first B rows = zero initial state, zeroed by the caller), logits [N,C],
 *   fed [N] int64 (embedding row used as input of each row), commits [N] int64 (label committed at each row)
 * Shapes as mh_hwcell_seq_fwd (H <= 512, H % 4 == 0, B <= 32); workspace >= mh_decoder_greedy_ws_bytes(N). */
size_t mh_decoder_greedy_ws_bytes(int N);
int mh_decoder_greedy(int H, int B, int T, const int *batch_sizes_host, int C, const float *enc_proj,
                      const float *emb_proj, const float *w_state, const float *b_state,
                      const float *dropout /*[B,H] or NULL*/, const float *w_out /*[C,H]*/, const float *b_out,
                      const long long *labels, float *h_buf, float *c_buf, float *logits, long long *fed,
                      long long *commits, void *workspace, size_t ws_bytes, void *stream);

/* Class-wise greedy suppression of the decoder's commitments in SGDet evaluation (lib/lstm/decoder_rnn.py:230-247, which
 * moves an [N,N,C] IoU tensor and the probabilities to the host): probs [N,C] = softmax of the class logits, boxes [N,C,4]
 * = the class-specific boxes; N rounds of {first arg-max of the table, commit its class, zero that class for every box
 * whose class box overlaps (IoU >= thresh, +1 convention, operation order of box_utils.nms_overlaps), retire the row};
 * commits [N] int64.  One workgroup, N*C*4 bytes of LDS (N*C <= 38400). */
size_t mh_decoder_nms_commit_max_bytes(void);   /* largest N*C*4 table the kernel holds in LDS on the current device (0: query failed) */
int mh_decoder_nms_commit(const float *probs, const float *boxes, int N, int C, float thresh, long long *commits,
                          void *stream);

/* Device-side fault state of the persistent kernels above.
 *   mh_fault_pending(): number of devices whose fault word is set (0 = none); host read, never synchronises.
 *   mh_fault_clear()  : re-arm the entry points after the caller has discarded the affected results.
 *   mh_debug_lstm_barrier_fault(enable): TEST HOOK -- while enabled, persistent launches run with an unreachable
 *                       barrier target, so the time-out path (fault word, NaN poisoning, MH_EFAULT) can be exercised. */
int mh_fault_pending(void);
int mh_fault_clear(void);
int mh_debug_lstm_barrier_fault(int enable);

/* ---------------------------------------------------------------------------------------------
 * Tail of the training step: global grad-norm clip + SGD(momentum, weight decay) as multi-tensor kernels.
 * Replaces lib/pytorch_misc.py:416-455 (`clip_grad_norm`: one host sync per parameter) and torch.optim.SGD
 * (models/train_rels.py:57-72,143-150).
 *   chunks: device array of 32-byte records {float* p; const float* g; float* buf; int32 n; float lr}, each
 *           covering at most mh_opt_chunk_elems() consecutive elements of one parameter (built once).
 *   mh_multi_sumsq  : sumsq_out[0] = sum over all chunks of g^2 (device scalar; partial = nchunks floats scratch)
 *   mh_multi_sgd_step: g' = g * min(1, max_norm/(sqrt(sumsq)+1e-6)) (skipped when sumsq == NULL or max_norm <= 0);
 *                      d = g' + wd*p; buf = first_step ? d : momentum*buf + d; p -= lr*buf
 *                      DEVICE-SIDE GUARD: with clipping on, a step whose sumsq is NaN / inf (gradients poisoned by a
 *                      timed-out persistent launch, an overflow) is skipped by the kernel itself -- weights and momentum
 *                      untouched -- and counted in a host-pinned word: the host may be several steps ahead and could not
 *                      have stopped it.  mh_opt_skipped_steps(): total skipped steps so far (host read, no sync);
 *                      mh_opt_skipped_clear() resets the counter.
 * ------------------------------------------------------------------------------------------- */
int mh_opt_chunk_elems(void);
int mh_opt_skipped_steps(void);
int mh_opt_skipped_clear(void);
/* Expand a HOST list of nparams records {p, g, buf, n = elements of the whole parameter, lr} (same 32-byte layout) into the
 * device chunk table chunks[nchunks], nchunks = sum over parameters of ceil(n / mh_opt_chunk_elems()).  The list travels in
 * the kernel arguments (no host->device copy, nothing read from params_host after the call returns). */
int mh_opt_build_chunks(const void *params_host, int nparams, void *chunks, int nchunks, void *stream);
int mh_multi_sumsq(const void *chunks, int nchunks, float *partial, float *sumsq_out, void *stream);
int mh_multi_sgd_step(const void *chunks, int nchunks, const float *sumsq, float max_norm, float momentum,
                      float weight_decay, int first_step, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused BatchNorm(train) / ReLU-mask / max-pool kernels of the union-box mask tower (NHWC fp32;
 * lib/get_union_boxes.py:31-39 Conv->ReLU->BN->MaxPool(3,2,1)->Conv->ReLU->BN, + RoIAligned features).
 *   mh_bn_stats        : per-channel mean / invstd of x[M,C] (biased var + eps), running stats updated (momentum,
 *                        unbiased var) when running_mean/var are given -- nn.BatchNorm semantics
 *   mh_bn_pool_fwd     : z = maxpool3x3/2/pad1( gamma*(x-mean)*invstd + beta ), argmax byte (ky*3+kx) per output
 *   mh_bn_residual_nchw: out[n,c,p] = residual[n,c,p] + BN(x)[n,p,c]  (x NHWC [N,P,C] -> NCHW, residual may be NULL)
 *   mh_nchw_to_nhwc_small: [N,C,P] -> [N,P,C]
 *   mh_bn_bwd          : dx (with the producer's ReLU mask x>0 when relu_mask), dgamma, dbeta from the dense
 *                        gradient g[N,H,W,C] or, when pooled, from the pooled gradient g[N,H/2,W/2,C] + argmax
 * ------------------------------------------------------------------------------------------- */
size_t mh_bn_ws_bytes(long long M, int C);
int mh_bn_stats(const float *x, long long M, int C, float eps, float momentum, float *mean, float *invstd,
                float *running_mean, float *running_var, void *workspace, size_t ws_bytes, void *stream);
int mh_bn_pool_fwd(const float *x, long long N, int H, int W, int C, const float *mean, const float *invstd,
                   const float *gamma, const float *beta, float *z, unsigned char *argmax, void *stream);
int mh_bn_residual_nchw(const float *x, long long N, int P, int C, const float *mean, const float *invstd,
                        const float *gamma, const float *beta, const float *residual_nchw, float *out_nchw,
                        void *stream);
/* y = act(BN(x) + residual) on NHWC rows [M,C] (ResNet bottleneck epilogues, lib/resnet.py:25-46;
 * mean/invstd from mh_bn_stats in train mode or from the running statistics in eval mode) */
int mh_bn_apply_nhwc(const float *x, long long M, int C, const float *mean, const float *invstd,
                     const float *gamma, const float *beta, const float *residual /*or NULL*/, int relu,
                     float *out, void *stream);
int mh_nchw_to_nhwc_small(const float *in_nchw, long long N, int P, int C, float *out_nhwc, void *stream);
int mh_bn_bwd(const float *x, const float *g, const unsigned char *argmax, long long N, int H, int W, int C,
              const float *mean, const float *invstd, const float *gamma, int pooled, int relu_mask, float *dx,
              float *dgamma, float *dbeta, void *workspace, size_t ws_bytes, void *stream);

/* The tower's FIRST convolution, direct (replaces nn.Conv2d(2, dim/2, kernel_size=7, stride=2, padding=3) + ReLU of
 * /root/reference lib/get_union_boxes.py:31-32 on the [N,S,S,2] NHWC masks of draw_union_boxes, S = 27); no column matrix.
 * Round 6: on the matrix cores -- every lane builds its MFMA fragments straight from the padded masks (a kernel row = 14
 * contiguous floats = one 16-deep k-tile); forward f16x3 (masks scaled by 2^14, per-channel weight exponents), weight gradient
 * bf16x6 with one output row per k-tile; fp32 products as everywhere.  MH_TOWER_CONV1=valu (environment, read per
 * call) selects round 4's kernels: thread = output channel with its 98 weights in registers, mask values through the scalar
 * cache, exact fp32 FMAs.
 *   mh_tower_conv1_out_size      : output height = width for mask size S (27 -> 14)
 *   mh_tower_conv1_padded_bytes  : bytes of the zero-padded mask copy [N, S+6, S+6, 2] (kept for the weight gradient)
 *   mh_tower_conv1_pad           : rects [N,S,S,2] -> padded
 *   mh_tower_conv1_fwd           : y[N,Ho,Wo,C0] = relu(conv(padded, w) + bias); w_kc[98][C0], k = (ky*7 + kx)*2 + ci
 *   mh_tower_conv1_wgrad         : dw_kc[99][C0]: rows 0..97 = weight gradient in w_kc's layout, row 98 = bias gradient
 *                                  (sum of dy), from dy[N,Ho,Wo,C0]; deterministic two-stage sum; workspace >= _ws_bytes
 * C0 a multiple of 256; Ho even and <= 16.  MH_ERR_BAD_ARG otherwise (the caller then uses mh_im2col_nhwc + mh_gemm_f32). */
int mh_tower_conv1_out_size(int S);
size_t mh_tower_conv1_padded_bytes(long long N, int S);
size_t mh_tower_conv1_wgrad_ws_bytes(long long N, int C0);
int mh_tower_conv1_pad(const float *rects_nhwc, long long N, int S, float *padded, void *stream);
int mh_tower_conv1_fwd(const float *padded, long long N, int S, const float *w_kc, const float *bias, int C0, float *y_nhwc,
                       void *stream);
int mh_tower_conv1_wgrad(const float *padded, const float *dy_nhwc, long long N, int S, int C0, float *dw_kc, void *workspace,
                         size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MOTIFS_HIP_H */
