// Detection post-processing of the evaluation call in seven launches, no host read inside.
//
// Replaces the torch-op chain of [D2-upstream] FastRCNNOutputLayers.inference as the reference reaches it
// (ovr/modeling/roi_heads/roi_emb_heads.py:280,357 -> predict_boxes / fast_rcnn_inference): Box2BoxTransform.apply_deltas,
// Boxes.clip, `scores > SCORE_THRESH_TEST` over the K foreground columns, class-wise NMS (batched_nms: boxes shifted by
// class * (max coordinate + 1), greedy in descending score order, IoU > NMS_THRESH_TEST suppressed) and the top
// DETECTIONS_PER_IMAGE in descending score order.  The reference evaluates with TEST.IMS_PER_BATCH 1 (configs/coco_stt.yaml:50):
// one image x 1000 proposals x 1204 columns per call, where the chain's ~100 small launches, four host reads and a
// sequential 3 000-step NMS sweep cost 1.8 ms behind a 3.0 ms head.
//
//   det_decode_clip_kernel   a thread per proposal: apply_deltas + clip, step by step as the torch ops round (this file is
//                            compiled with -ffp-contract=off; `deltas / w` is torch's multiplication by 1 / w)
//   det_count_kernel         a wave per proposal: how many of its K class probabilities pass the threshold
//   det_scan_kernel          one workgroup: per image, the exclusive scan of those counts (= where each proposal's candidates go)
//   det_emit_kernel          a wave per proposal: its candidates as 64-bit keys  class | ~score bits | row  in (row, class) order
//   det_sort_kernel          a workgroup per image: bitonic sort of the keys in LDS (class-major, descending score, ties by row =
//                            candidate order); shifted boxes, class starts
//   det_pairs_kernel         a thread per (candidate, 64 predecessors): which earlier candidates of its class overlap it (IoU on the
//                            SHIFTED boxes, so every decision rounds as batched_nms's does) -- the m^2 / 2 pair tests of a
//                            crowded class, chip-wide, into per-candidate overlap bit sets
//   det_nms_topk_kernel      a workgroup per image: the greedy sweep as a fix-point over kept / undecided bit sets; second sort of
//                            the survivors by (descending score, row, class) = the reference's order; top-k out.
//
// Candidate order, tie-breaking and every fp32 operation equal the torch chain's (tests/test_gpu_postprocess.py: bit-identical
// detections).  What the kernels cannot take is flagged on the device and read by the caller with the counts (its ONE host
// read): non-finite boxes / scores (the reference drops such proposals with a warning) and an image with more candidates than
// the LDS sort holds -- the caller then runs the torch chain.
#include "common.h"

namespace locov {

constexpr int kDetMaxCand = LOCOV_DETECT_MAX_CANDIDATES;       // candidates per image (64-bit keys sorted in LDS)
constexpr int kDetThreads = 1024;
constexpr int kRowBits = 14, kScoreBits = 32, kClsBits = 15;  // key = class << 46 | ~score << 14 | row

struct DetGeom {
    int n_img;
    int roff[LOCOV_LABEL_MAX_IMAGES + 1];                       // proposals of image i: rows [roff[i], roff[i + 1])
    float h[LOCOV_LABEL_MAX_IMAGES], w[LOCOV_LABEL_MAX_IMAGES];
};

__device__ __forceinline__ int det_image_of(const DetGeom &g, int r)
{
    int lo = 0, hi = g.n_img - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (g.roff[mid] <= r) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ bool det_finite(float v) { return fabsf(v) <= 3.402823466e38f; }        // false for NaN / inf

// Box2BoxTransform.apply_deltas (class-agnostic: one box per proposal) + Boxes.clip, rounded as the torch ops round
__global__ __launch_bounds__(256) void det_decode_clip_kernel(const float4 *__restrict__ deltas, const float4 *__restrict__ props, int R,
                                                              DetGeom g, float inv_wx, float inv_wy, float inv_ww, float inv_wh,
                                                              float scale_clamp, float4 *__restrict__ boxes, int *__restrict__ flags)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float4 d = deltas[r], b = props[r];
    const float widths = __fsub_rn(b.z, b.x), heights = __fsub_rn(b.w, b.y);
    const float ctr_x = __fadd_rn(b.x, __fmul_rn(0.5f, widths)), ctr_y = __fadd_rn(b.y, __fmul_rn(0.5f, heights));
    const float dx = __fmul_rn(d.x, inv_wx), dy = __fmul_rn(d.y, inv_wy);
    float dw = __fmul_rn(d.z, inv_ww), dh = __fmul_rn(d.w, inv_wh);
    dw = dw > scale_clamp ? scale_clamp : dw;                     // torch.clamp(max=): NaN stays NaN
    dh = dh > scale_clamp ? scale_clamp : dh;
    const float pcx = __fadd_rn(__fmul_rn(dx, widths), ctr_x), pcy = __fadd_rn(__fmul_rn(dy, heights), ctr_y);
    const float pw = __fmul_rn(expf(dw), widths), ph = __fmul_rn(expf(dh), heights);
    float4 o;
    o.x = __fsub_rn(pcx, __fmul_rn(0.5f, pw));
    o.y = __fsub_rn(pcy, __fmul_rn(0.5f, ph));
    o.z = __fadd_rn(pcx, __fmul_rn(0.5f, pw));
    o.w = __fadd_rn(pcy, __fmul_rn(0.5f, ph));
    if (!(det_finite(o.x) && det_finite(o.y) && det_finite(o.z) && det_finite(o.w))) atomicOr(flags, LOCOV_DETECT_FLAG_NONFINITE);
    const int img = det_image_of(g, r);
    const float W = g.w[img], H = g.h[img];
    o.x = fminf(fmaxf(o.x, 0.f), W);
    o.y = fminf(fmaxf(o.y, 0.f), H);
    o.z = fminf(fmaxf(o.z, 0.f), W);
    o.w = fminf(fmaxf(o.w, 0.f), H);
    boxes[r] = o;
}

// a wave per proposal: candidates (p > thr among the K foreground columns); every one of the K + 1 columns must be finite
__global__ __launch_bounds__(256) void det_count_kernel(const float *__restrict__ probs, int64_t ld, int K, int R, float thr,
                                                        int *__restrict__ row_count, int *__restrict__ flags)
{
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float *p = probs + (int64_t)r * ld;
    int n = 0;
    bool bad = false;
    for (int c = lane; c <= K; c += 64) {
        const float v = p[c];
        bad |= !det_finite(v);
        n += (c < K && v > thr) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (__ballot(bad) != 0ull && lane == 0) atomicOr(flags, LOCOV_DETECT_FLAG_NONFINITE);
    if (lane == 0) row_count[r] = n;
}

// one workgroup: row_off[r] = candidates of the rows of r's image in front of r; img_count[i] = candidates of image i
__global__ __launch_bounds__(kDetThreads) void det_scan_kernel(const int *__restrict__ row_count, DetGeom g, int *__restrict__ row_off,
                                                               int *__restrict__ img_count)
{
    __shared__ int wave_sum[kDetThreads / 64];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int img = 0; img < g.n_img; img++) {
        const int r0 = g.roff[img], r1 = g.roff[img + 1];
        if (tid == 0) carry = 0;
        __syncthreads();
        for (int base = r0; base < r1; base += kDetThreads) {
            const int r = base + tid;
            const int v = r < r1 ? row_count[r] : 0;
            int incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            if (lane == 63) wave_sum[wave] = incl;
            __syncthreads();
            int before = carry;
            for (int w = 0; w < wave; w++) before += wave_sum[w];
            if (r < r1) row_off[r] = before + incl - v;
            __syncthreads();
            if (tid == kDetThreads - 1) carry = before + incl;
            __syncthreads();
        }
        if (tid == 0) img_count[img] = carry;
        __syncthreads();
    }
}

// a wave per proposal: its candidates' keys, in class order, behind the candidates of the image's earlier rows
__global__ __launch_bounds__(256) void det_emit_kernel(const float *__restrict__ probs, int64_t ld, int K, int R, float thr, DetGeom g,
                                                       const int *__restrict__ row_off, unsigned long long *__restrict__ keys)
{
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int img = det_image_of(g, r);
    const float *p = probs + (int64_t)r * ld;
    unsigned long long *out = keys + (int64_t)img * kDetMaxCand;
    const unsigned long long row = (unsigned long long)(r - g.roff[img]);
    int at = row_off[r];
    for (int c0 = 0; c0 < K; c0 += 64) {
        const int c = c0 + lane;
        const float v = c < K ? p[c] : 0.f;
        const bool ok = c < K && v > thr;
        const unsigned long long b = __ballot(ok);
        if (ok) {
            const int pos = at + __popcll(b & ((1ull << lane) - 1ull));
            if (pos < kDetMaxCand)
                out[pos] = ((unsigned long long)c << (kRowBits + kScoreBits)) | ((unsigned long long)(~__float_as_uint(v)) << kRowBits) | row;
        }
        at += __popcll(b);
    }
}

__device__ __forceinline__ bool det_iou_gt(const float4 a, const float4 b, float thr)        // (= nms.hip's iou_gt)
{
    const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    const float sa = (a.z - a.x) * (a.w - a.y), sb = (b.z - b.x) * (b.w - b.y);
    return inter / (sa + sb - inter) > thr;
}

__device__ __forceinline__ void det_bitonic_sort(unsigned long long *key, int P, int tid)
{
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (P >> 1); t += kDetThreads) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool ascending = (lo & size) == 0;
                const unsigned long long a = key[lo], b = key[hi];
                if ((a > b) == ascending) {
                    key[lo] = b;
                    key[hi] = a;
                }
            }
            __syncthreads();
        }
}

// ---- selection: sort (a workgroup per image) -> pair tests (the whole chip) -> NMS fix-point + top-k (a workgroup per image) -----------
//
// Classes never suppress each other (that is all batched_nms's coordinate shift achieves; the IoU is still taken on the SHIFTED
// boxes so that every decision rounds as the reference's does), so the greedy sweep runs inside the class segments of the sorted
// candidate list.  A class with m candidates needs m (m - 1) / 2 pair tests -- 500 000 for a class every proposal of an image is
// a candidate of -- which ONE compute unit takes 0.8 ms over; spread over the chip they take microseconds.  So the tests go to
// their own launch, which leaves for every candidate the bit set "which earlier candidates of my class overlap me", and the sweep
// becomes a fix-point over two bit sets (kept / undecided) that ends in exactly the sequential sweep's result.
constexpr int kDetOvWords = kDetMaxCand / 128 * kDetMaxCand + kDetMaxCand;     // 64-bit words of overlap bit sets per image, worst case

struct DetLists {                       // per-image work lists in the workspace (element offsets are per image: * kDetMaxCand etc.)
    unsigned long long *keys;           // [n_img][kDetMaxCand]        candidates; sorted in place by det_sort_kernel
    int *cstart;                        // [n_img][kDetMaxCand]        first candidate of candidate i's class
    int *woff;                          // [n_img][kDetMaxCand + 4]    where candidate i's overlap words start; [n] = their total
    float4 *cbox;                       // [n_img][kDetMaxCand]        candidate i's shifted box
    unsigned long long *ov;             // [n_img][kDetOvWords]        bit j of word w of candidate i: candidate cstart + 64 w + j overlaps it
};

__global__ __launch_bounds__(kDetThreads) void det_sort_kernel(DetLists L, const int *__restrict__ img_count, const float4 *__restrict__ boxes,
                                                               DetGeom g, int *__restrict__ flags)
{
    extern __shared__ unsigned long long key[];                  // P keys
    __shared__ float red[kDetThreads / 64];
    __shared__ int wave_sum[kDetThreads / 64];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = img_count[img];
    if (n > kDetMaxCand) {
        if (tid == 0) atomicOr(flags, LOCOV_DETECT_FLAG_OVERFLOW);
        return;
    }
    if (n == 0) return;
    int P = 2;
    while (P < n) P <<= 1;
    unsigned long long *keys_g = L.keys + (int64_t)img * kDetMaxCand;
    for (int i = tid; i < P; i += kDetThreads) key[i] = i < n ? keys_g[i] : ~0ull;
    __syncthreads();
    det_bitonic_sort(key, P, tid);                               // class-major; inside a class: descending score, ties by row
    constexpr unsigned long long kRowMask = (1ull << kRowBits) - 1ull;
    const float4 *gbox = boxes + g.roff[img];
    // batched_nms's coordinate offset: class * (max coordinate of the image's candidate boxes + 1)
    float m = -__builtin_inff();
    for (int i = tid; i < n; i += kDetThreads) {
        const float4 b = gbox[(int)(key[i] & kRowMask)];
        m = fmaxf(m, fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
    for (int w = 1; w < kDetThreads / 64; w++) m = fmaxf(m, red[w]);
    const float shift_unit = m + 1.f;
    int *cstart = L.cstart + (int64_t)img * kDetMaxCand, *woff = L.woff + (int64_t)img * (kDetMaxCand + 4);
    float4 *cbox = L.cbox + (int64_t)img * kDetMaxCand;
    // sorted keys, shifted boxes, class starts; and -- a block-wide exclusive scan, candidates in CONSECUTIVE runs of kQ per thread --
    // where each candidate's overlap words go
    constexpr int kQ = kDetMaxCand / kDetThreads;
    int words[kQ], run = 0;
#pragma unroll
    for (int q = 0; q < kQ; q++) {
        const int i = tid * kQ + q;
        words[q] = 0;
        if (i < n) {
            const unsigned long long k = key[i];
            const int cls = (int)(k >> (kRowBits + kScoreBits));
            const unsigned long long first_of_class = (unsigned long long)cls << (kRowBits + kScoreBits);
            int lo = 0, hi = i;
            while (lo < hi) {                                    // the first candidate of i's class (the keys are sorted)
                const int mid = (lo + hi) >> 1;
                if (key[mid] < first_of_class) lo = mid + 1;
                else hi = mid;
            }
            float4 b = gbox[(int)(k & kRowMask)];
            const float off = (float)cls * shift_unit;
            b.x += off;
            b.y += off;
            b.z += off;
            b.w += off;
            keys_g[i] = k;
            cbox[i] = b;
            cstart[i] = lo;
            words[q] = (i - lo + 63) >> 6;
        }
        run += words[q];
    }
    int incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int before = incl - run;
    for (int w = 0; w < wave; w++) before += wave_sum[w];
#pragma unroll
    for (int q = 0; q < kQ; q++) {
        const int i = tid * kQ + q;
        if (i < n) woff[i] = before;
        before += words[q];
    }
    if (tid == kDetThreads - 1) woff[n] = before;                // (thread 1023 holds the tail: every i >= n contributes no word)
}

// a thread per (candidate, 64 predecessors): which earlier candidates of its class overlap it (IoU > threshold on the shifted boxes).
// grid (candidates / 256, images, word index): the word index runs to the worst case, the workgroups past an image's longest class
// leave at once
__global__ __launch_bounds__(256) void det_pairs_kernel(DetLists L, const int *__restrict__ img_count, float nms_thr)
{
    const int img = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x, w = blockIdx.z;
    const int n = img_count[img];
    if (n > kDetMaxCand || a >= n) return;
    const int s = L.cstart[(int64_t)img * kDetMaxCand + a];
    const int base = s + 64 * w;
    if (base >= a) return;
    const float4 *cbox = L.cbox + (int64_t)img * kDetMaxCand;
    const float4 box_a = cbox[a];
    const int lim = min(64, a - base);
    unsigned long long bits = 0;
#pragma unroll 8
    for (int j = 0; j < lim; j++) bits |= (unsigned long long)det_iou_gt(cbox[base + j], box_a, nms_thr) << j;
    L.ov[(int64_t)img * kDetOvWords + L.woff[(int64_t)img * (kDetMaxCand + 4) + a] + w] = bits;
}

__device__ __forceinline__ unsigned long long det_window64(const unsigned *bits, int pos)      // 64 bits of a bit set from bit `pos` on
{
    const int w = pos >> 5, sh = pos & 31;
    const unsigned long long lo = (unsigned long long)bits[w] | ((unsigned long long)bits[w + 1] << 32);
    return sh ? (lo >> sh) | ((unsigned long long)bits[w + 2] << (64 - sh)) : lo;
}

__global__ __launch_bounds__(kDetThreads) void det_nms_topk_kernel(DetLists L, const int *__restrict__ img_count, const float4 *__restrict__ boxes,
                                                                   DetGeom g, int topk, float4 *__restrict__ out_boxes, float *__restrict__ out_scores,
                                                                   int64_t *__restrict__ out_classes, int64_t *__restrict__ out_rows,
                                                                   int *__restrict__ counts)
{
    extern __shared__ unsigned long long key[];                  // P keys (the second sort's)
    __shared__ unsigned kept[kDetMaxCand / 32 + 4], und[kDetMaxCand / 32 + 4];
    __shared__ int n_keep_s;
    const int img = blockIdx.x, tid = threadIdx.x;
    const int n = img_count[img];
    if (n > kDetMaxCand || n == 0) {
        if (tid == 0) counts[img] = 0;
        return;
    }
    int P = 2;
    while (P < n) P <<= 1;
    const unsigned long long *keys_g = L.keys + (int64_t)img * kDetMaxCand;
    const int *cstart = L.cstart + (int64_t)img * kDetMaxCand, *woff = L.woff + (int64_t)img * (kDetMaxCand + 4);
    const unsigned long long *ov = L.ov + (int64_t)img * kDetOvWords;
    for (int w = tid; w < kDetMaxCand / 32 + 4; w += kDetThreads) {
        kept[w] = 0;
        const int lo = w * 32;
        und[w] = lo + 32 <= n ? 0xffffffffu : (lo < n ? (1u << (n - lo)) - 1u : 0u);
    }
    if (tid == 0) n_keep_s = 0;
    __syncthreads();
    // Greedy NMS as a fix-point over the two bit sets: an undecided candidate with a KEPT overlapping predecessor is suppressed;
    // with none kept and none undecided it is kept; otherwise it waits.  A candidate is marked kept BEFORE it leaves the undecided
    // set and readers look at the undecided set FIRST, so a reader never takes a just-decided candidate for a suppressed one;
    // bits only ever move one way, so a stale read costs a round, never the result (= the sequential sweep's).
    constexpr int kQ = kDetMaxCand / kDetThreads;
    for (;;) {
        int pending = 0;
#pragma unroll
        for (int q = 0; q < kQ; q++) {
            const int i = tid + q * kDetThreads;
            if (i < n && ((und[i >> 5] >> (i & 31)) & 1u)) {
                const int s = cstart[i];
                const unsigned long long *mine = ov + woff[i];
                bool sup = false, wait = false;
                for (int w = 0, base = s; base < i && !sup; w++, base += 64) {
                    const unsigned long long mask = mine[w];
                    if (!mask) continue;
                    const unsigned long long u = det_window64(und, base);     // (undecided first, then kept: see above)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const unsigned long long k = det_window64(kept, base);
                    if (mask & k) sup = true;
                    else if (mask & u) wait = true;
                }
                if (sup) atomicAnd(&und[i >> 5], ~(1u << (i & 31)));
                else if (!wait) {
                    atomicOr(&kept[i >> 5], 1u << (i & 31));
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    atomicAnd(&und[i >> 5], ~(1u << (i & 31)));
                } else
                    pending = 1;
            }
        }
        if (__syncthreads_or(pending) == 0) break;
    }
    // the survivors in the reference's order: descending score, ties in candidate (row, class) order
    constexpr unsigned long long kRowMask = (1ull << kRowBits) - 1ull;
    int mine_n = 0;
    for (int i = tid; i < P; i += kDetThreads) {
        unsigned long long v = ~0ull;
        if (i < n && ((kept[i >> 5] >> (i & 31)) & 1u)) {
            const unsigned long long k = keys_g[i];
            const unsigned long long cls = k >> (kRowBits + kScoreBits), nscore = (k >> kRowBits) & 0xffffffffull, row = k & kRowMask;
            v = (nscore << (kRowBits + kClsBits)) | (row << kClsBits) | cls;
            mine_n++;
        }
        key[i] = v;
    }
    if (mine_n) atomicAdd(&n_keep_s, mine_n);
    __syncthreads();
    det_bitonic_sort(key, P, tid);
    const int n_keep = n_keep_s;
    const int count = n_keep < topk ? n_keep : topk;
    const float4 *gbox = boxes + g.roff[img];
    for (int j = tid; j < count; j += kDetThreads) {
        const unsigned long long k = key[j];
        const int cls = (int)(k & ((1ull << kClsBits) - 1ull)), row = (int)((k >> kClsBits) & kRowMask);
        const unsigned nscore = (unsigned)(k >> (kRowBits + kClsBits));
        const int64_t slot = (int64_t)img * topk + j;
        out_boxes[slot] = gbox[row];
        out_scores[slot] = __uint_as_float(~nscore);
        out_classes[slot] = cls;
        out_rows[slot] = row;
    }
    if (tid == 0) counts[img] = count;
}

}  // namespace locov

using namespace locov;

extern "C" {

static int64_t det_align16(int64_t v) { return (v + 15) & ~(int64_t)15; }

int64_t locov_detect_postprocess_workspace_bytes(int64_t R, int n_images)
{
    if (R <= 0 || n_images <= 0) return 0;
    // clipped boxes [R, 4] fp32, row counts [R], row offsets [R], candidates per image [n_images]; then per image: keys, class starts,
    // overlap-word offsets, shifted boxes (kDetMaxCand each) and the overlap bit sets (kDetOvWords 64-bit words: 4.3 MB)
    const int64_t head = det_align16(R * 24 + (int64_t)((n_images + 3) & ~3) * 4);
    const int64_t per_image = (int64_t)kDetMaxCand * (8 + 4 + 16) + (int64_t)(kDetMaxCand + 4) * 4 + (int64_t)kDetOvWords * 8;
    return head + (int64_t)n_images * per_image + 64;
}

int locov_detect_postprocess(const float *probs, int64_t ld_probs, int num_classes, const float *deltas, const float *proposal_boxes,
                             const int *row_offsets, const float *image_hw, int n_images, float wx, float wy, float ww, float wh,
                             float scale_clamp, float score_thresh, float nms_thresh, int topk, void *workspace, int64_t workspace_bytes,
                             float *out_boxes, float *out_scores, int64_t *out_classes, int64_t *out_rows, int *counts_and_flags,
                             locov_stream_t stream)
{
    LOCOV_REQUIRE(n_images >= 0 && n_images <= LOCOV_LABEL_MAX_IMAGES, "locov_detect_postprocess: 0..%d images per call", LOCOV_LABEL_MAX_IMAGES);
    if (n_images == 0) return LOCOV_OK;
    LOCOV_REQUIRE(row_offsets && image_hw, "locov_detect_postprocess: null host array");
    LOCOV_REQUIRE(num_classes >= 1 && num_classes < (1 << kClsBits), "locov_detect_postprocess: 1..%d classes", (1 << kClsBits) - 1);
    LOCOV_REQUIRE(topk >= 1 && topk <= kDetMaxCand, "locov_detect_postprocess: 1 <= topk <= %d", kDetMaxCand);
    LOCOV_REQUIRE(ld_probs >= (int64_t)num_classes + 1, "locov_detect_postprocess: ld_probs must cover the K + 1 columns");
    LOCOV_REQUIRE(wx != 0.f && wy != 0.f && ww != 0.f && wh != 0.f, "locov_detect_postprocess: zero box weight");
    DetGeom g{};
    g.n_img = n_images;
    for (int i = 0; i <= n_images; i++) {
        g.roff[i] = row_offsets[i];
        LOCOV_REQUIRE(g.roff[i] >= 0 && (i == 0 || g.roff[i] >= g.roff[i - 1]), "locov_detect_postprocess: offsets must be non-decreasing");
        LOCOV_REQUIRE(i == 0 || g.roff[i] - g.roff[i - 1] < (1 << kRowBits), "locov_detect_postprocess: at most %d proposals per image",
                      (1 << kRowBits) - 1);
    }
    LOCOV_REQUIRE(g.roff[0] == 0, "locov_detect_postprocess: offsets start at 0");
    for (int i = 0; i < n_images; i++) {
        g.h[i] = image_hw[2 * i];
        g.w[i] = image_hw[2 * i + 1];
    }
    const int64_t R = g.roff[n_images];
    LOCOV_REQUIRE(counts_and_flags, "locov_detect_postprocess: null pointer");
    LOCOV_REQUIRE(R == 0 || (probs && deltas && proposal_boxes && workspace && out_boxes && out_scores && out_classes && out_rows),
                  "locov_detect_postprocess: null pointer");
    LOCOV_REQUIRE(R == 0 || workspace_bytes >= locov_detect_postprocess_workspace_bytes(R, n_images), "locov_detect_postprocess: workspace too small");
    LOCOV_REQUIRE(((uintptr_t)deltas | (uintptr_t)proposal_boxes | (uintptr_t)workspace | (uintptr_t)out_boxes) % 16 == 0,
                  "locov_detect_postprocess: boxes / workspace must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    hipError_t e = hipMemsetAsync(counts_and_flags, 0, sizeof(int) * (size_t)(n_images + 1), s);
    if (e != hipSuccess) return set_error(LOCOV_ERR_LAUNCH, "locov_detect_postprocess: memset: %s", hipGetErrorString(e));
    if (R == 0) return LOCOV_OK;
    char *ws = static_cast<char *>(workspace);
    float4 *boxes = reinterpret_cast<float4 *>(ws);
    int *row_count = reinterpret_cast<int *>(ws + R * 16);
    int *row_off = row_count + R;
    int *img_count = row_off + R;
    char *lists = ws + det_align16(R * 24 + (int64_t)((n_images + 3) & ~3) * 4);
    DetLists L;
    L.keys = reinterpret_cast<unsigned long long *>(lists);
    L.cbox = reinterpret_cast<float4 *>(lists + (int64_t)n_images * kDetMaxCand * 8);
    L.ov = reinterpret_cast<unsigned long long *>(lists + (int64_t)n_images * kDetMaxCand * 24);
    L.cstart = reinterpret_cast<int *>(lists + (int64_t)n_images * kDetMaxCand * 24 + (int64_t)n_images * kDetOvWords * 8);
    L.woff = L.cstart + (int64_t)n_images * kDetMaxCand;
    unsigned long long *keys = L.keys;
    int *flags = counts_and_flags + n_images;
    // (torch divides a tensor by a python scalar by multiplying with the reciprocal formed in fp32)
    const float inv_wx = 1.0f / wx, inv_wy = 1.0f / wy, inv_ww = 1.0f / ww, inv_wh = 1.0f / wh;
    hipLaunchKernelGGL(det_decode_clip_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, s, reinterpret_cast<const float4 *>(deltas),
                       reinterpret_cast<const float4 *>(proposal_boxes), (int)R, g, inv_wx, inv_wy, inv_ww, inv_wh, scale_clamp, boxes, flags);
    hipLaunchKernelGGL(det_count_kernel, dim3((unsigned)ceil_div(R, 4)), dim3(256), 0, s, probs, ld_probs, num_classes, (int)R, score_thresh,
                       row_count, flags);
    hipLaunchKernelGGL(det_scan_kernel, dim3(1), dim3(kDetThreads), 0, s, row_count, g, row_off, img_count);
    hipLaunchKernelGGL(det_emit_kernel, dim3((unsigned)ceil_div(R, 4)), dim3(256), 0, s, probs, ld_probs, num_classes, (int)R, score_thresh, g,
                       row_off, keys);
    // (the sorts' LDS: 8 bytes per candidate slot, 64 KB; the attribute belongs to the kernel on ONE device)
    const size_t lds = (size_t)kDetMaxCand * 8;
    static int attr_state[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set_error(LOCOV_ERR_LAUNCH, "locov_detect_postprocess: hipGetDevice");
    if (attr_state[dev] == 0)
        attr_state[dev] = (hipFuncSetAttribute(reinterpret_cast<const void *>(det_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds) == hipSuccess &&
                           hipFuncSetAttribute(reinterpret_cast<const void *>(det_nms_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds) == hipSuccess) ? 1 : -1;
    if (attr_state[dev] != 1) return set_error(LOCOV_ERR_LAUNCH, "locov_detect_postprocess: cannot raise the dynamic LDS limit to %zu bytes", lds);
    hipLaunchKernelGGL(det_sort_kernel, dim3((unsigned)n_images), dim3(kDetThreads), lds, s, L, img_count, boxes, g, flags);
    // (a class holds at most one candidate per proposal: its predecessors fit ceil(rows of the largest image / 64) words)
    int max_rows = 0;
    for (int i = 0; i < n_images; i++) max_rows = g.roff[i + 1] - g.roff[i] > max_rows ? g.roff[i + 1] - g.roff[i] : max_rows;
    const int pair_words = (int)ceil_div(max_rows < kDetMaxCand ? max_rows : kDetMaxCand, 64);
    hipLaunchKernelGGL(det_pairs_kernel, dim3(kDetMaxCand / 256, (unsigned)n_images, (unsigned)pair_words), dim3(256), 0, s, L, img_count,
                       nms_thresh);
    hipLaunchKernelGGL(det_nms_topk_kernel, dim3((unsigned)n_images), dim3(kDetThreads), lds, s, L, img_count, boxes, g, topk,
                       reinterpret_cast<float4 *>(out_boxes), out_scores, out_classes, out_rows, counts_and_flags);
    return check_launch("locov_detect_postprocess");
}

}  // extern "C"
