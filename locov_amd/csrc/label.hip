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

// The sampler behind the labelling, for a batch whose every image fills its budget (the training forwards' speculation,
// roi_emb_heads.py:25-118 -> [D2-upstream] ROIHeads._sample_proposals / subsample_labels): image i takes its
// num_pos = min(foreground candidates, max_pos) foreground proposals of smallest key_pos, then budget - num_pos background proposals
// of smallest key_neg, in key order -- the first entries of the image's segment of argsort(key_pos) / argsort(key_neg), which is
// what the torch form reads off two global sorts -- and gathers every field of the sampled Instances in the same launch.
// grid (image, 2): workgroup (i, 0) sorts image i by key_pos and fills the foreground slots, (i, 1) by key_neg and fills the rest.
// The sort is a bitonic network over the image's keys in LDS (at most kSampleMax proposals per image), ties broken by row number.
constexpr int kSampleMax = 4096, kSampleThreads = 1024;

struct SampleGeom {
    int n_img;
    int roff[LOCOV_LABEL_MAX_IMAGES + 1];
    int goff[LOCOV_LABEL_MAX_IMAGES + 1];
    int budget, max_pos;
};

__global__ __launch_bounds__(kSampleThreads) void sample_proposals_kernel(
    const double *__restrict__ key_pos, const double *__restrict__ key_neg, const int64_t *__restrict__ labels,
    const int64_t *__restrict__ gt_index, const unsigned long long *__restrict__ rows, const float4 *__restrict__ boxes,
    const float4 *__restrict__ gt, const float *__restrict__ field, SampleGeom g, int64_t num_classes, int64_t *__restrict__ picked,
    float4 *__restrict__ out_boxes, int64_t *__restrict__ out_classes, float4 *__restrict__ out_gt, int64_t *__restrict__ out_fg,
    float *__restrict__ rois, float *__restrict__ field_out)
{
    __shared__ double k[kSampleMax];
    __shared__ unsigned short idx[kSampleMax];
    const int img = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    const int r0 = g.roff[img], n = g.roff[img + 1] - r0;
    if (n <= 0) return;
    int P = 1;
    while (P < n) P <<= 1;
    const double *key = (part == 0 ? key_pos : key_neg) + r0;
    for (int i = tid; i < P; i += kSampleThreads) {
        k[i] = i < n ? key[i] : __builtin_inf();
        idx[i] = (unsigned short)i;
    }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (P >> 1); t += kSampleThreads) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool ascending = (lo & size) == 0;
                const double a = k[lo], b = k[hi];
                const unsigned short ia = idx[lo], ib = idx[hi];
                const bool a_after_b = a > b || (a == b && ia > ib);
                if (a_after_b == ascending) {
                    k[lo] = b;
                    k[hi] = a;
                    idx[lo] = ib;
                    idx[hi] = ia;
                }
            }
            __syncthreads();
        }
    const unsigned long long avail = rows[4 * img];
    const unsigned long long population = part == 0 ? avail : rows[4 * img + 1];      // candidates of THIS part's kind
    const int num_pos = avail < (unsigned long long)g.max_pos ? (int)avail : g.max_pos;
    const int j_lo = part == 0 ? 0 : num_pos, j_hi = part == 0 ? num_pos : g.budget;
    const bool has_gt = g.goff[img + 1] > g.goff[img];
    for (int j = j_lo + tid; j < j_hi && j < g.budget; j += kSampleThreads) {
        // (an image that does not fill its budget reads past its population: the caller throws such a batch's sample away, but
        //  it has already enqueued the losses on it -- the index has to stay inside the image, and the class has to be one a
        //  loss can take: a row past the population may carry the IGNORE label -1 of a matcher with an ignore band, which
        //  torch's cross-entropy answers with a device-side assert; such a slot counts as background)
        const int rank = j - j_lo;
        const int row = r0 + (int)idx[rank < n ? rank : n - 1];
        const int64_t slot = (int64_t)img * g.budget + j;
        const float4 b = boxes[row];
        const int64_t cls = (unsigned long long)rank < population ? labels[row] : num_classes;
        picked[slot] = row;
        out_boxes[slot] = b;
        out_classes[slot] = cls;
        out_fg[slot] = cls != num_classes ? 1 : 0;
        out_gt[slot] = has_gt ? gt[gt_index[row]] : float4{0.f, 0.f, 0.f, 0.f};
        float *r = rois + 5 * slot;
        r[0] = (float)img;
        r[1] = b.x;
        r[2] = b.y;
        r[3] = b.z;
        r[4] = b.w;
        if (field_out != nullptr) field_out[slot] = field[row];
    }
}

}  // namespace locov

extern "C" int locov_sample_proposals(const double *key_pos, const double *key_neg, const int64_t *labels, const int64_t *gt_index,
                                      const int64_t *rows, const float *boxes, const float *gt_boxes, const float *field,
                                      const int *prop_offsets, const int *gt_offsets, int n_images, int budget, int max_pos,
                                      int64_t num_classes, int64_t *picked, float *out_boxes, int64_t *out_classes, float *out_gt_boxes,
                                      int64_t *out_fg, float *rois, float *field_out, locov_stream_t stream)
{
    using namespace locov;
    LOCOV_REQUIRE(n_images >= 0 && n_images <= LOCOV_LABEL_MAX_IMAGES, "locov_sample_proposals: 0..%d images per call", LOCOV_LABEL_MAX_IMAGES);
    LOCOV_REQUIRE(budget >= 0 && max_pos >= 0 && max_pos <= budget, "locov_sample_proposals: 0 <= max_pos <= budget");
    if (n_images == 0 || budget == 0) return LOCOV_OK;
    LOCOV_REQUIRE(prop_offsets && gt_offsets, "locov_sample_proposals: null host array");
    SampleGeom g{};
    g.n_img = n_images;
    g.budget = budget;
    g.max_pos = max_pos;
    for (int i = 0; i <= n_images; i++) {
        g.roff[i] = prop_offsets[i];
        g.goff[i] = gt_offsets[i];
        LOCOV_REQUIRE(g.roff[i] >= 0 && g.goff[i] >= 0 && (i == 0 || (g.roff[i] >= g.roff[i - 1] && g.goff[i] >= g.goff[i - 1])),
                      "locov_sample_proposals: offsets must be non-decreasing");
        LOCOV_REQUIRE(i == 0 || (g.roff[i] - g.roff[i - 1] >= 1 && g.roff[i] - g.roff[i - 1] <= kSampleMax),
                      "locov_sample_proposals: 1..%d proposals per image (got %d)", kSampleMax, i ? g.roff[i] - g.roff[i - 1] : 0);
    }
    LOCOV_REQUIRE(g.roff[0] == 0 && g.goff[0] == 0, "locov_sample_proposals: offsets start at 0");
    LOCOV_REQUIRE(key_pos && key_neg && labels && gt_index && rows && boxes && picked && out_boxes && out_classes && out_gt_boxes && out_fg &&
                      rois && (g.goff[n_images] == 0 || gt_boxes) && ((field == nullptr) == (field_out == nullptr)),
                  "locov_sample_proposals: null pointer");
    LOCOV_REQUIRE(((uintptr_t)boxes | (uintptr_t)gt_boxes | (uintptr_t)out_boxes | (uintptr_t)out_gt_boxes) % 16 == 0,
                  "locov_sample_proposals: boxes must be 16-byte aligned");
    hipLaunchKernelGGL(sample_proposals_kernel, dim3((unsigned)n_images, 2), dim3(kSampleThreads), 0, as_stream(stream), key_pos, key_neg, labels,
                       gt_index, reinterpret_cast<const unsigned long long *>(rows), reinterpret_cast<const float4 *>(boxes),
                       reinterpret_cast<const float4 *>(gt_boxes), field, g, num_classes, picked, reinterpret_cast<float4 *>(out_boxes),
                       out_classes, reinterpret_cast<float4 *>(out_gt_boxes), out_fg, rois, field_out);
    return check_launch("locov_sample_proposals");
}

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
