/*
 * oracle/roi_ops_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar fp32, no FMA contraction: build with
 * -ffp-contract=off) of the third-party arithmetic the LocOV ROI head reaches
 * through Detectron2's ROIPooler:
 *
 *   reference call site : ovr/modeling/roi_heads/roi_emb_heads.py:182-187 (pooler
 *                         construction: output_size=14, scales=(1/16,),
 *                         sampling_ratio=0, pooler_type="ROIAlignV2")
 *                         ovr/modeling/roi_heads/roi_emb_heads.py:243-245
 *                         (_shared_roi_transform -> self.pooler(features, boxes))
 *   algorithm lives in  : detectron2 (unpinned; README.md:23-30 "follow install
 *                         instructions", era v0.6) ROIPooler / ROIAlign ->
 *                         torchvision.ops.roi_align (unpinned, ~0.11).  Neither is
 *                         vendored under /root/reference nor installed, so the
 *                         published algorithm is restated here.
 *
 * PARITY STATUS: "parity unpinned" w.r.t. reference-held vectors -- the reference
 * has no tests or fixtures.  Pinned instead by analytic cases in
 * tests/test_oracle_roi_align.py (constant map, linear ramp, out-of-range box,
 * aligned half-pixel shift) and by a torch-op cross-check of the level formula.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product path (locov_amd/) never does.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* One pre-computed bilinear sample: 4 plane offsets + 4 weights
 * (torchvision roi_align CPU kernel: pre_calc_for_bilinear_interpolate). */
typedef struct {
    int pos1, pos2, pos3, pos4;
    float w1, w2, w3, w4;
} oracle_sample_t;

static void oracle_precalc(int height, int width, int pooled_h, int pooled_w,
                           float roi_start_h, float roi_start_w, float bin_h,
                           float bin_w, int grid_h, int grid_w,
                           oracle_sample_t *tab)
{
    int idx = 0;
    for (int ph = 0; ph < pooled_h; ph++) {
        for (int pw = 0; pw < pooled_w; pw++) {
            for (int iy = 0; iy < grid_h; iy++) {
                const float yy = roi_start_h + (float)ph * bin_h +
                                 ((float)iy + .5f) * bin_h / (float)grid_h;
                for (int ix = 0; ix < grid_w; ix++) {
                    const float xx = roi_start_w + (float)pw * bin_w +
                                     ((float)ix + .5f) * bin_w / (float)grid_w;
                    float x = xx, y = yy;
                    oracle_sample_t s;
                    /* outside [-1, H] x [-1, W]: contributes zero */
                    if (y < -1.0f || y > (float)height || x < -1.0f ||
                        x > (float)width) {
                        memset(&s, 0, sizeof(s));
                        tab[idx++] = s;
                        continue;
                    }
                    if (y <= 0) y = 0;
                    if (x <= 0) x = 0;
                    int y_low = (int)y, x_low = (int)x, y_high, x_high;
                    if (y_low >= height - 1) {
                        y_high = y_low = height - 1;
                        y = (float)y_low;
                    } else {
                        y_high = y_low + 1;
                    }
                    if (x_low >= width - 1) {
                        x_high = x_low = width - 1;
                        x = (float)x_low;
                    } else {
                        x_high = x_low + 1;
                    }
                    const float ly = y - (float)y_low, lx = x - (float)x_low;
                    const float hy = 1.f - ly, hx = 1.f - lx;
                    s.w1 = hy * hx; s.w2 = hy * lx; s.w3 = ly * hx; s.w4 = ly * lx;
                    s.pos1 = y_low * width + x_low;
                    s.pos2 = y_low * width + x_high;
                    s.pos3 = y_high * width + x_low;
                    s.pos4 = y_high * width + x_high;
                    tab[idx++] = s;
                }
            }
        }
    }
}

/*
 * torchvision.ops.roi_align forward, NCHW fp32.
 *   feat [N,C,H,W], rois [R,5] = (batch_idx, x0, y0, x1, y1), out [R,C,ph,pw].
 * `aligned` = ROIAlignV2 half-pixel shift (Detectron2 ROIAlign(aligned=True)).
 * sampling_ratio <= 0  -> adaptive grid ceil(roi_size / pooled_size).
 * Returns 0, or -1 on a bad batch index / allocation failure.
 */
int oracle_roi_align_fwd(const float *feat, int N, int C, int H, int W,
                         const float *rois, int64_t R, int pooled_h,
                         int pooled_w, float spatial_scale, int sampling_ratio,
                         int aligned, float *out)
{
    int status = 0;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t n = 0; n < R; n++) {
        const float *roi = rois + n * 5;
        const int b = (int)roi[0];
        float *o = out + n * (int64_t)C * pooled_h * pooled_w;
        if (b < 0 || b >= N) {
            status = -1;
            continue;
        }
        const float offset = aligned ? 0.5f : 0.0f;
        const float roi_start_w = roi[1] * spatial_scale - offset;
        const float roi_start_h = roi[2] * spatial_scale - offset;
        const float roi_end_w = roi[3] * spatial_scale - offset;
        const float roi_end_h = roi[4] * spatial_scale - offset;
        float roi_width = roi_end_w - roi_start_w;
        float roi_height = roi_end_h - roi_start_h;
        if (!aligned) { /* legacy: force malformed ROIs to be 1x1 */
            roi_width = roi_width > 1.f ? roi_width : 1.f;
            roi_height = roi_height > 1.f ? roi_height : 1.f;
        }
        const float bin_h = roi_height / (float)pooled_h;
        const float bin_w = roi_width / (float)pooled_w;
        int grid_h = sampling_ratio > 0 ? sampling_ratio
                                        : (int)ceilf(roi_height / (float)pooled_h);
        int grid_w = sampling_ratio > 0 ? sampling_ratio
                                        : (int)ceilf(roi_width / (float)pooled_w);
        const int gprod = grid_h * grid_w;
        const float count = (float)(gprod > 1 ? gprod : 1);
        if (grid_h < 0) grid_h = 0;
        if (grid_w < 0) grid_w = 0;
        const size_t ntab = (size_t)pooled_h * pooled_w * grid_h * grid_w;
        oracle_sample_t *tab =
            (oracle_sample_t *)malloc((ntab ? ntab : 1) * sizeof(oracle_sample_t));
        if (!tab) {
            status = -1;
            continue;
        }
        oracle_precalc(H, W, pooled_h, pooled_w, roi_start_h, roi_start_w, bin_h,
                       bin_w, grid_h, grid_w, tab);
        for (int c = 0; c < C; c++) {
            const float *plane = feat + ((int64_t)b * C + c) * H * W;
            size_t idx = 0;
            for (int ph = 0; ph < pooled_h; ph++) {
                for (int pw = 0; pw < pooled_w; pw++) {
                    float v = 0.f;
                    for (int iy = 0; iy < grid_h; iy++) {
                        for (int ix = 0; ix < grid_w; ix++) {
                            const oracle_sample_t s = tab[idx++];
                            v += s.w1 * plane[s.pos1] + s.w2 * plane[s.pos2] +
                                 s.w3 * plane[s.pos3] + s.w4 * plane[s.pos4];
                        }
                    }
                    v /= count;
                    o[((int64_t)c * pooled_h + ph) * pooled_w + pw] = v;
                }
            }
        }
        free(tab);
    }
    return status;
}

/*
 * ROIAlign backward (torchvision roi_align_backward): scatters
 * grad_out[R,C,ph,pw] * w / count to the 4 taps of every sample.
 * grad_feat [N,C,H,W] must be zeroed by the caller.  Serial (deterministic).
 */
int oracle_roi_align_bwd(const float *grad_out, int N, int C, int H, int W,
                         const float *rois, int64_t R, int pooled_h,
                         int pooled_w, float spatial_scale, int sampling_ratio,
                         int aligned, float *grad_feat)
{
    for (int64_t n = 0; n < R; n++) {
        const float *roi = rois + n * 5;
        const int b = (int)roi[0];
        if (b < 0 || b >= N) return -1;
        const float offset = aligned ? 0.5f : 0.0f;
        const float roi_start_w = roi[1] * spatial_scale - offset;
        const float roi_start_h = roi[2] * spatial_scale - offset;
        const float roi_end_w = roi[3] * spatial_scale - offset;
        const float roi_end_h = roi[4] * spatial_scale - offset;
        float roi_width = roi_end_w - roi_start_w;
        float roi_height = roi_end_h - roi_start_h;
        if (!aligned) {
            roi_width = roi_width > 1.f ? roi_width : 1.f;
            roi_height = roi_height > 1.f ? roi_height : 1.f;
        }
        const float bin_h = roi_height / (float)pooled_h;
        const float bin_w = roi_width / (float)pooled_w;
        int grid_h = sampling_ratio > 0 ? sampling_ratio
                                        : (int)ceilf(roi_height / (float)pooled_h);
        int grid_w = sampling_ratio > 0 ? sampling_ratio
                                        : (int)ceilf(roi_width / (float)pooled_w);
        const int gprod = grid_h * grid_w;
        const float count = (float)(gprod > 1 ? gprod : 1);
        if (grid_h < 0) grid_h = 0;
        if (grid_w < 0) grid_w = 0;
        const size_t ntab = (size_t)pooled_h * pooled_w * grid_h * grid_w;
        oracle_sample_t *tab =
            (oracle_sample_t *)malloc((ntab ? ntab : 1) * sizeof(oracle_sample_t));
        if (!tab) return -1;
        oracle_precalc(H, W, pooled_h, pooled_w, roi_start_h, roi_start_w, bin_h,
                       bin_w, grid_h, grid_w, tab);
        for (int c = 0; c < C; c++) {
            float *plane = grad_feat + ((int64_t)b * C + c) * H * W;
            const float *g = grad_out + (n * C + c) * (int64_t)pooled_h * pooled_w;
            size_t idx = 0;
            for (int ph = 0; ph < pooled_h; ph++) {
                for (int pw = 0; pw < pooled_w; pw++) {
                    const float gv = g[ph * pooled_w + pw];
                    for (int iy = 0; iy < grid_h; iy++) {
                        for (int ix = 0; ix < grid_w; ix++) {
                            const oracle_sample_t s = tab[idx++];
                            plane[s.pos1] += gv * s.w1 / count;
                            plane[s.pos2] += gv * s.w2 / count;
                            plane[s.pos3] += gv * s.w3 / count;
                            plane[s.pos4] += gv * s.w4 / count;
                        }
                    }
                }
            }
        }
        free(tab);
    }
    return 0;
}

/*
 * Detectron2 poolers.assign_boxes_to_levels:
 *   size  = sqrt(area),  area = (x1-x0)*(y1-y0)              (fp32)
 *   lvl   = floor(canonical_level + log2(size/canonical_box_size + 1e-8))  (fp32)
 *   lvl   = clamp(lvl, min_level, max_level) - min_level      (int64)
 * log2 is evaluated in double and rounded once to fp32 (= a correctly rounded
 * log2f); torch's own CPU log2f is accurate to <=1 ULP, so the two agree except
 * within 1 ULP of a level boundary -- tests/test_oracle_levels.py cross-checks
 * against the same formula written with torch ops.
 */
void oracle_level_assign(const float *boxes, int64_t R, int min_level,
                         int max_level, int canonical_box_size,
                         int canonical_level, int64_t *out)
{
    for (int64_t i = 0; i < R; i++) {
        const float *b = boxes + i * 4;
        const float area = (b[2] - b[0]) * (b[3] - b[1]);
        const float size = sqrtf(area);
        const float arg = size / (float)canonical_box_size + 1e-8f;
        const float l2 = (float)log2((double)arg);
        float lvl = floorf((float)canonical_level + l2);
        if (!(lvl >= (float)min_level)) lvl = (float)min_level; /* NaN -> min */
        if (lvl > (float)max_level) lvl = (float)max_level;
        out[i] = (int64_t)lvl - min_level;
    }
}

/* Spatial mean over HW of an [R,C,HW] tensor: sequential fp32 sum / HW
 * (reference: roi_emb_heads.py:262,344,356  box_features.mean(dim=[2,3])). */
void oracle_spatial_mean(const float *x, int64_t R, int C, int HW, float *out)
{
#pragma omp parallel for
    for (int64_t i = 0; i < R * (int64_t)C; i++) {
        const float *p = x + i * HW;
        float s = 0.f;
        for (int k = 0; k < HW; k++) s += p[k];
        out[i] = s / (float)HW;
    }
}

/* y[M,N] = x[M,K] . W[N,K]^T + b  (nn.Linear; box_emb_head.py:196,206,211).
 * Double accumulation: this is the "exact" reference the 1e-4 fp32 logit gate is
 * measured against. */
void oracle_linear(const float *x, int64_t M, int K, const float *W,
                   const float *bias, int N, float *y)
{
#pragma omp parallel for
    for (int64_t m = 0; m < M; m++) {
        for (int n = 0; n < N; n++) {
            double acc = 0.0;
            const float *xr = x + m * K, *wr = W + (int64_t)n * K;
            for (int k = 0; k < K; k++) acc += (double)xr[k] * (double)wr[k];
            if (bias) acc += (double)bias[n];
            y[m * N + n] = (float)acc;
        }
    }
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
