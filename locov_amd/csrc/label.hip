// Proposal labelling of a whole batch in ONE launch: IoU against the image's ground-truth boxes, [D2-upstream] Matcher thresholds,
// class labels, the sampler's sort keys, per-image population counts and the two validity bits the reference asserts on the host --
// the device half of SampleAllROIHeads.label_and_sample_proposals (ovr/modeling/roi_heads/roi_emb_heads.py:25-118: pairwise_iou ->
// proposal_matcher -> _sample_proposals' labelling; [D2-upstream] subsample_labels draws from the two populations).
//
// One thread per proposal walks ITS image's ground-truth boxes (a handful): what the torch form does with ~80 small launches per
// batch (a quarter of a training step's launch count) and a [sum M, sum R] matrix.  Every arithmetic step is the torch op it
// replaces, rounded individually (this file is built with -ffp-contract=off): the IoU values, hence the argmax (first maximum, as
// torch.max), the threshold tests and the labels are the torch form's, bit for bit.
#include "common.h"

namespace locov {

struct LabelGeom {
    int n_img;
    int roff[LOCOV_LABEL_MAX_IMAGES + 1];    // proposals of image i: rows [roff[i], roff[i+1]) of the concatenated boxes
    int goff[LOCOV_LABEL_MAX_IMAGES + 1];    // its ground truth: rows [goff[i], goff[i+1])
    int n_thr;                               // Matcher: n_thr intervals [lo[k], hi[k]) with label lab[k] in {-1, 0, 1}
    float lo[LOCOV_LABEL_MAX_THRESHOLDS], hi[LOCOV_LABEL_MAX_THRESHOLDS];
    int lab[LOCOV_LABEL_MAX_THRESHOLDS];
};

__global__ __launch_bounds__(256) void label_proposals_kernel(const float4 *__restrict__ boxes, const float4 *__restrict__ gt,
                                                              const int64_t *__restrict__ gt_classes, LabelGeom g, int64_t num_classes,
                                                              const double *__restrict__ rnd, int64_t *__restrict__ gt_index,
                                                              int64_t *__restrict__ labels, double *__restrict__ key_pos,
                                                              double *__restrict__ key_neg, unsigned long long *__restrict__ rows)
{
    const int total = g.roff[g.n_img];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int img = 0;
    while (i >= g.roff[img + 1]) img++;                   // (a few images: a short scalar walk)
    const float4 b = boxes[i];
    const float bw = b.z - b.x, bh = b.w - b.y;
    const float area_b = bw * bh;
    // matched_vals, matches = quality.max(dim=0) over this image's rows; rows of other images carry -1 in the torch form and can
    // never win, an image without ground truth leaves (-1, row 0)
    float best = -1.f;
    int best_j = 0;
    bool bad = false;
    for (int j = g.goff[img]; j < g.goff[img + 1]; j++) {
        const float4 a = gt[j];
        const float area_a = (a.z - a.x) * (a.w - a.y);
        float w = fminf(a.z, b.z) - fmaxf(a.x, b.x), h = fminf(a.w, b.w) - fmaxf(a.y, b.y);
        // torch.min / torch.max propagate NaN (fminf / fmaxf do not); clamp_(min=0) keeps it
        if (a.z != a.z || b.z != b.z || a.x != a.x || b.x != b.x) w = __builtin_nanf("");
        if (a.w != a.w || b.w != b.w || a.y != a.y || b.y != b.y) h = __builtin_nanf("");
        w = w < 0.f ? 0.f : w;
        h = h < 0.f ? 0.f : h;
        const float inter = w * h;
        const float q = inter > 0.f ? inter / ((area_a + area_b) - inter) : 0.f;
        bad |= !(q >= 0.f);
        const bool first = j == g.goff[img];
        if (first || q > best || (q != q && best == best)) {          // strict >: the FIRST maximum; NaN counts as the maximum
            best = q;
            best_j = j;
        }
    }
    int ml = 1;                                            // Matcher: match_labels start at 1, every interval that holds overwrites
    for (int k = 0; k < g.n_thr; k++)
        if (best >= g.lo[k] && best < g.hi[k]) ml = g.lab[k];
    int64_t label;
    if (g.goff[img + 1] == g.goff[img])
        label = num_classes;                               // an image without ground truth: ROIHeads._sample_proposals' has_gt == False
                                                           // branch labels EVERY proposal background, whatever the Matcher said
                                                           // (with IOU_LABELS[0] == -1 the threshold label of "no match" is "ignore")
    else
        label = ml == 0 ? num_classes : ml == -1 ? (int64_t)-1 : gt_classes[best_j];
    const bool pos = label != -1 && label != num_classes, neg = label == num_classes;
    gt_index[i] = best_j;
    labels[i] = label;
    const double base = (double)img * 4.0;
    key_pos[i] = (rnd[i] + (pos ? 0.0 : 2.0)) + base;
    key_neg[i] = (rnd[total + i] + (neg ? 0.0 : 2.0)) + base;
    const bool degenerate = !((bw > 0.f) && (bh > 0.f)) && pos;
    unsigned long long *row = rows + 4 * img;
    if (pos) atomicAdd(row + 0, 1ull);
    if (neg) atomicAdd(row + 1, 1ull);
    if (bad) atomicAdd(row + 2, 1ull);
    if (degenerate) atomicAdd(row + 3, 1ull);
}

}  // namespace locov

extern "C" int locov_label_proposals(const float *boxes, const int *prop_offsets, const float *gt_boxes, const int64_t *gt_classes,
                                     const int *gt_offsets, int n_images, const float *thr_lo, const float *thr_hi, const int *thr_label,
                                     int n_thresholds, int64_t num_classes, const double *rnd, int64_t *gt_index, int64_t *labels,
                                     double *key_pos, double *key_neg, int64_t *rows, locov_stream_t stream)
{
    using namespace locov;
    LOCOV_REQUIRE(n_images >= 0 && n_images <= LOCOV_LABEL_MAX_IMAGES, "locov_label_proposals: 0..%d images per call", LOCOV_LABEL_MAX_IMAGES);
    LOCOV_REQUIRE(n_thresholds >= 0 && n_thresholds <= LOCOV_LABEL_MAX_THRESHOLDS, "locov_label_proposals: at most %d matcher intervals",
                  LOCOV_LABEL_MAX_THRESHOLDS);
    if (n_images == 0) return LOCOV_OK;
    LOCOV_REQUIRE(prop_offsets && gt_offsets && (n_thresholds == 0 || (thr_lo && thr_hi && thr_label)), "locov_label_proposals: null host array");
    LabelGeom g{};
    g.n_img = n_images;
    for (int i = 0; i <= n_images; i++) {
        g.roff[i] = prop_offsets[i];
        g.goff[i] = gt_offsets[i];
        LOCOV_REQUIRE(g.roff[i] >= 0 && g.goff[i] >= 0 && (i == 0 || (g.roff[i] >= g.roff[i - 1] && g.goff[i] >= g.goff[i - 1])),
                      "locov_label_proposals: offsets must be non-decreasing");
    }
    LOCOV_REQUIRE(g.roff[0] == 0 && g.goff[0] == 0, "locov_label_proposals: offsets start at 0");
    g.n_thr = n_thresholds;
    for (int k = 0; k < n_thresholds; k++) {
        g.lo[k] = thr_lo[k];
        g.hi[k] = thr_hi[k];
        g.lab[k] = thr_label[k];
    }
    const int total = g.roff[n_images];
    if (total == 0) return LOCOV_OK;
    LOCOV_REQUIRE(boxes && rnd && gt_index && labels && key_pos && key_neg && rows && (g.goff[n_images] == 0 || (gt_boxes && gt_classes)),
                  "locov_label_proposals: null pointer");
    LOCOV_REQUIRE(((uintptr_t)boxes | (uintptr_t)gt_boxes) % 16 == 0, "locov_label_proposals: boxes must be 16-byte aligned");
    hipLaunchKernelGGL(label_proposals_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(boxes), reinterpret_cast<const float4 *>(gt_boxes), gt_classes, g, num_classes, rnd,
                       gt_index, labels, key_pos, key_neg, reinterpret_cast<unsigned long long *>(rows));
    return check_launch("locov_label_proposals");
}
