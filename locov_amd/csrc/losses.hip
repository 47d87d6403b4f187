// The two scalar-sized loss tails of a training step, each as ONE launch instead of a chain of ~40 small torch launches (and as many
// again in autograd's backward): what a step of the LSM / STT configurations spends on them is launch gaps, not work.
//
//   * locov_box_reg_loss -- [D2-upstream] FastRCNNOutputLayers.box_reg_loss as the reference's heads call it
//     (ovr/modeling/roi_heads/box_emb_grounding_head.py:278-279,370-374: smooth_l1, beta 0 by default): Box2BoxTransform.get_deltas
//     of (proposal, matched ground truth) for the foreground rows, smooth-L1 against the predicted deltas, summed and divided by the
//     number of ALL rows; the gradient with respect to the predictions comes out of the same launch.
//   * locov_grounding_ce_fwd / _bwd -- the cross-entropy tail of GroundingHead.forward (ovr/modeling/mmss_heads/grounding_head.py:
//     239-251 the "(max + 100)" replacement of pairs with neither words nor regions, :273-290 log_softmax over captions and over
//     images + the diagonal means, :357-377 the batch accuracies) on the [B, B] caption x image cost matrices of locov_grounding_fwd.
//
// Built with -ffp-contract=off: each step is the torch op it replaces, rounded on its own; the sums run in a fixed order (one
// workgroup, a fixed tree), so a step's losses are reproducible run to run.
#include "common.h"

namespace locov {

namespace {

constexpr int kLossThreads = 256;

// fixed-order sum over the workgroup (every thread returns the total)
__device__ __forceinline__ float block_sum(float v, float *red)
{
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int s = kLossThreads / 2; s > 0; s >>= 1) {
        if (t < s) red[t] = red[t] + red[t + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ float block_max(float v, float *red)
{
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int s = kLossThreads / 2; s > 0; s >>= 1) {
        if (t < s) red[t] = fmaxf(red[t], red[t + s]);
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

}  // namespace

__global__ __launch_bounds__(kLossThreads) void box_reg_loss_kernel(const float4 *__restrict__ src, const float4 *__restrict__ tgt,
                                                                    const float *__restrict__ pred, int64_t ld,
                                                                    const int64_t *__restrict__ cls, int64_t R, int64_t num_classes,
                                                                    float wx, float wy, float ww, float wh, float beta,
                                                                    float *__restrict__ loss, float *__restrict__ dpred)
{
    __shared__ float red[kLossThreads];
    const bool agnostic = ld == 4;
    const float n = (float)(R > 1 ? R : 1);
    float part = 0.f;
    for (int64_t r = threadIdx.x; r < R; r += kLossThreads) {
        const int64_t c = cls[r];
        const bool fg = c >= 0 && c < num_classes;
        const int64_t col = agnostic ? 0 : (c < 0 ? 0 : (c >= num_classes ? num_classes - 1 : c)) * 4;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        if (fg) {
            const float4 s = src[r], t = tgt[r];
            // Box2BoxTransform.get_deltas, op by op
            const float sw = s.z - s.x, sh = s.w - s.y;
            const float scx = s.x + 0.5f * sw, scy = s.y + 0.5f * sh;
            const float tw = t.z - t.x, th = t.w - t.y;
            const float tcx = t.x + 0.5f * tw, tcy = t.y + 0.5f * th;
            const float d[4] = {wx * (tcx - scx) / sw, wy * (tcy - scy) / sh, ww * logf(tw / sw), wh * logf(th / sh)};
            const float *p = pred + r * ld + col;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float e = p[j] - d[j], a = fabsf(e);
                float l, de;                                  // fvcore smooth_l1_loss (beta < 1e-5: plain L1) and its derivative in e
                if (beta < 1e-5f) {
                    l = a;
                    de = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
                } else if (a < beta) {
                    l = 0.5f * (a * a) / beta;
                    de = e / beta;
                } else {
                    l = a - 0.5f * beta;
                    de = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
                }
                part = part + l;
                g[j] = de / n;
            }
        }
        if (dpred) {
            // class-agnostic: the whole [R, 4] gradient is written here; per-class predictions: the caller zeroed [R, 4 K] and the
            // four columns of the row's class are filled in (background / ignored rows: nothing, as the indexed upstream form)
            if (agnostic || fg) {
                float *q = dpred + r * ld + col;
#pragma unroll
                for (int j = 0; j < 4; j++) q[j] = g[j];
            }
        }
    }
    const float total = block_sum(part, red);
    if (threadIdx.x == 0) loss[0] = total / n;
}

// out[tag * 4 + k], tag 0 = cost0 ("Words": w2r), 1 = cost1 ("Regions": r2w); k: 0 = CE choose caption (softmax over dim 0),
// 1 = CE choose image (dim 1), 2 / 3 = the batch accuracies of the same two directions.  GRAD: d(sum_k up[tag*2+k] * CE_k) / d cost.
template <bool GRAD>
__global__ __launch_bounds__(kLossThreads) void grounding_ce_kernel(const float *__restrict__ cost0, const float *__restrict__ cost1,
                                                                    const float *__restrict__ cmask, const float *__restrict__ rmask, int B,
                                                                    int T, int NR, float *__restrict__ out, const float *up0,
                                                                    const float *up1, const float *up2, const float *up3,
                                                                    float *__restrict__ d0, float *__restrict__ d1)
{
    __shared__ float red[kLossThreads];
    __shared__ float z[LOCOV_GROUNDING_CE_MAX_B * LOCOV_GROUNDING_CE_MAX_B];       // -cost' of the current tag
    __shared__ float nw[LOCOV_GROUNDING_CE_MAX_B], nr[LOCOV_GROUNDING_CE_MAX_B];
    __shared__ float cmx[LOCOV_GROUNDING_CE_MAX_B], cls_[LOCOV_GROUNDING_CE_MAX_B]; // per column: max, log-sum-exp term (softmax over dim 0)
    __shared__ float rmx[LOCOV_GROUNDING_CE_MAX_B], rls_[LOCOV_GROUNDING_CE_MAX_B]; // per row (softmax over dim 1)
    const int t = threadIdx.x, BB = B * B;
    for (int i = t; i < B; i += kLossThreads) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < T; k++) a = a + cmask[i * T + k];
        for (int k = 0; k < NR; k++) b = b + rmask[i * NR + k];
        nw[i] = a;
        nr[i] = b;
    }
    __syncthreads();
    for (int tag = 0; tag < 2; tag++) {
        const float *cost = tag ? cost1 : cost0;
        if (!cost) continue;                                  // (workgroup-uniform)
        float m = -INFINITY;
        for (int e = t; e < BB; e += kLossThreads) m = fmaxf(m, cost[e]);
        const float fill = block_max(m, red) + 100.0f;       // pairs with neither words nor regions: max + 100 (:239-251)
        for (int e = t; e < BB; e += kLossThreads) {
            const int i = e / B, j = e - i * B;
            const bool ok = nw[i] > 0.f || nr[j] > 0.f;
            z[e] = -(ok ? cost[e] : fill);
        }
        __syncthreads();
        // log_softmax's pieces: x - max - log(sum exp(x - max)), per column (dim 0) and per row (dim 1)
        for (int u = t; u < 2 * B; u += kLossThreads) {
            const bool col = u < B;
            const int k = col ? u : u - B;
            float mx = -INFINITY;
            for (int v = 0; v < B; v++) mx = fmaxf(mx, col ? z[v * B + k] : z[k * B + v]);
            float s = 0.f;
            for (int v = 0; v < B; v++) s = s + expf((col ? z[v * B + k] : z[k * B + v]) - mx);
            (col ? cmx : rmx)[k] = mx;
            (col ? cls_ : rls_)[k] = logf(s);
        }
        __syncthreads();
        if (!GRAD) {
            float lc = 0.f, li = 0.f, ac = 0.f, ai = 0.f;
            for (int k = t; k < B; k += kLossThreads) {
                const float zkk = z[k * B + k];
                lc = -((zkk - cmx[k]) - cls_[k]);
                li = -((zkk - rmx[k]) - rls_[k]);
                // argmin of cost' = argmax of z, first index on ties
                int bc = 0, bi = 0;
                for (int v = 1; v < B; v++) {
                    if (z[v * B + k] > z[bc * B + k]) bc = v;
                    if (z[k * B + v] > z[k * B + bi]) bi = v;
                }
                ac = bc == k ? 1.f : 0.f;
                ai = bi == k ? 1.f : 0.f;
            }
            // (B <= kLossThreads: one diagonal element per thread, so the four partials above are that element's values)
            const float slc = block_sum(lc, red), sli = block_sum(li, red), sac = block_sum(ac, red), sai = block_sum(ai, red);
            if (t == 0) {
                out[tag * 4 + 0] = slc / (float)B;
                out[tag * 4 + 1] = sli / (float)B;
                out[tag * 4 + 2] = sac / (float)B;
                out[tag * 4 + 3] = sai / (float)B;
            }
        } else {
            const float *pc = tag ? up2 : up0, *pi = tag ? up3 : up1;
            const float gc = (pc ? pc[0] : 0.f) / (float)B, gi = (pi ? pi[0] : 0.f) / (float)B;
            float *d = tag ? d1 : d0;
            for (int e = t; e < BB; e += kLossThreads) {
                const int i = e / B, j = e - i * B;
                const bool ok = nw[i] > 0.f || nr[j] > 0.f;
                const float pcol = expf((z[e] - cmx[j]) - cls_[j]), prow = expf((z[e] - rmx[i]) - rls_[i]);
                const float dz = gc * (pcol - (i == j ? 1.f : 0.f)) + gi * (prow - (i == j ? 1.f : 0.f));
                d[e] = ok ? -dz : 0.f;                        // z = -cost'; the replaced pairs are constants
            }
        }
        __syncthreads();
    }
}

}  // namespace locov

extern "C" int locov_box_reg_loss(const float *proposal_boxes, const float *gt_boxes, const float *pred_deltas, int64_t ld,
                                  const int64_t *gt_classes, int64_t R, int64_t num_classes, float wx, float wy, float ww, float wh,
                                  float smooth_l1_beta, float *loss, float *dpred, locov_stream_t stream)
{
    using namespace locov;
    LOCOV_REQUIRE(R >= 0 && num_classes >= 1 && (ld == 4 || ld == 4 * num_classes),
                  "locov_box_reg_loss: pred_deltas must be [R, 4] or [R, 4 * num_classes]");
    LOCOV_REQUIRE(loss && (R == 0 || (proposal_boxes && gt_boxes && pred_deltas && gt_classes)), "locov_box_reg_loss: null pointer");
    LOCOV_REQUIRE(((uintptr_t)proposal_boxes | (uintptr_t)gt_boxes) % 16 == 0, "locov_box_reg_loss: boxes must be 16-byte aligned");
    hipLaunchKernelGGL(box_reg_loss_kernel, dim3(1), dim3(kLossThreads), 0, as_stream(stream), reinterpret_cast<const float4 *>(proposal_boxes),
                       reinterpret_cast<const float4 *>(gt_boxes), pred_deltas, ld, gt_classes, R, num_classes, wx, wy, ww, wh, smooth_l1_beta,
                       loss, dpred);
    return check_launch("locov_box_reg_loss");
}

static int grounding_ce_args(const float *c0, const float *c1, const float *cm, const float *rm, int B, int T, int NR)
{
    using namespace locov;
    LOCOV_REQUIRE(B >= 1 && B <= LOCOV_GROUNDING_CE_MAX_B && T >= 0 && NR >= 0, "locov_grounding_ce: 1 <= B <= %d", LOCOV_GROUNDING_CE_MAX_B);
    LOCOV_REQUIRE((c0 || c1) && cm && rm, "locov_grounding_ce: null pointer");
    return LOCOV_OK;
}

extern "C" int locov_grounding_ce_fwd(const float *cost_w2r, const float *cost_r2w, const float *caption_mask, const float *region_mask, int B,
                                      int T, int NR, float *out8, locov_stream_t stream)
{
    using namespace locov;
    if (int rc = grounding_ce_args(cost_w2r, cost_r2w, caption_mask, region_mask, B, T, NR)) return rc;
    LOCOV_REQUIRE(out8, "locov_grounding_ce_fwd: null output");
    hipLaunchKernelGGL(grounding_ce_kernel<false>, dim3(1), dim3(kLossThreads), 0, as_stream(stream), cost_w2r, cost_r2w, caption_mask,
                       region_mask, B, T, NR, out8, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    return check_launch("locov_grounding_ce_fwd");
}

extern "C" int locov_grounding_ce_bwd(const float *cost_w2r, const float *cost_r2w, const float *caption_mask, const float *region_mask, int B,
                                      int T, int NR, const float *g_w2r_caption, const float *g_w2r_image, const float *g_r2w_caption,
                                      const float *g_r2w_image, float *dcost_w2r, float *dcost_r2w, locov_stream_t stream)
{
    using namespace locov;
    if (int rc = grounding_ce_args(cost_w2r, cost_r2w, caption_mask, region_mask, B, T, NR)) return rc;
    LOCOV_REQUIRE((!cost_w2r || dcost_w2r) && (!cost_r2w || dcost_r2w), "locov_grounding_ce_bwd: null gradient output");
    hipLaunchKernelGGL(grounding_ce_kernel<true>, dim3(1), dim3(kLossThreads), 0, as_stream(stream), cost_w2r, cost_r2w, caption_mask,
                       region_mask, B, T, NR, nullptr, g_w2r_caption, g_w2r_image, g_r2w_caption, g_r2w_image, dcost_w2r, dcost_r2w);
    return check_launch("locov_grounding_ce_bwd");
}
