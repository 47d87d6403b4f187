// Greedy non-maximum suppression on the device (no host round trip).
//
// Replaces [D2-upstream] torchvision.ops.nms as reached from box_predictor.inference ->
// fast_rcnn_inference -> batched_nms (ovr/modeling/roi_heads/roi_emb_heads.py:280,357; SURVEY.md
// 8a-10).  torchvision is not available on the ROCm box; its semantics are kept: boxes visited in
// descending score order, a box is dropped when its IoU with an already kept box is > threshold,
// IoU = inter / (area_a + area_b - inter) on XYXY boxes.
//
// Two kernels: (1) a KxK "who suppresses whom" bit matrix, 64x64 box pairs per workgroup with the
// column boxes staged in LDS; (2) one workgroup sweeps the rows in score order, keeping the
// removed-set as K/64 words in LDS (a box's row is OR-ed in only if the box survives).  The sweep
// is inherently sequential in K but stays on the GPU: K = a few thousand candidates -> a few ms,
// and nothing is copied to the host.
#include "common.h"

namespace locov {

__device__ __forceinline__ bool iou_gt(const float4 a, const float4 b, float thr)
{
    const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    const float sa = (a.z - a.x) * (a.w - a.y), sb = (b.z - b.x) * (b.w - b.y);
    return inter / (sa + sb - inter) > thr;
}

// boxes are already sorted by descending score.  mask[i][cb] bit j = box (cb*64+j) is suppressed by box i.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4 *__restrict__ boxes, int K, float thr,
                                                      unsigned long long *__restrict__ mask, int col_blocks)
{
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;                                  // only later (lower-score) boxes can be suppressed
    __shared__ float4 cols[64];
    const int cbase = cb * 64, t = threadIdx.x;
    if (cbase + t < K) cols[t] = boxes[cbase + t];
    __syncthreads();
    const int i = rb * 64 + t;
    if (i >= K) return;
    const float4 me = boxes[i];
    const int ncols = min(64, K - cbase);
    unsigned long long bits = 0;
    for (int j = (rb == cb) ? t + 1 : 0; j < ncols; j++)
        if (iou_gt(me, cols[j], thr)) bits |= 1ull << j;
    mask[(int64_t)i * col_blocks + cb] = bits;
}

__global__ __launch_bounds__(256) void nms_sweep_kernel(const unsigned long long *__restrict__ mask, int K,
                                                        int col_blocks, unsigned char *__restrict__ keep,
                                                        int *__restrict__ num_keep)
{
    extern __shared__ unsigned long long removed[];
    for (int w = threadIdx.x; w < col_blocks; w += blockDim.x) removed[w] = 0;
    __syncthreads();
    int kept = 0;
    for (int i = 0; i < K; i++) {
        const bool alive = !((removed[i >> 6] >> (i & 63)) & 1ull);     // uniform: every thread reads the same word
        if (alive) {
            kept++;
            // rows only carry bits for columns >= their own block
            for (int w = (i >> 6) + threadIdx.x; w < col_blocks; w += blockDim.x)
                removed[w] |= mask[(int64_t)i * col_blocks + w];
        }
        if (threadIdx.x == 0) keep[i] = alive ? 1 : 0;
        __syncthreads();
    }
    if (threadIdx.x == 0) *num_keep = kept;
}

}  // namespace locov

using namespace locov;

extern "C" {

// workspace: K * ceil(K/64) 64-bit words (caller-allocated; locov_nms_workspace_bytes)
int64_t locov_nms_workspace_bytes(int64_t K) { return K <= 0 ? 0 : K * ((K + 63) / 64) * 8; }

int locov_nms_sorted(const float *boxes_sorted, int64_t K, float iou_threshold, void *workspace,
                     unsigned char *keep, int *num_keep, locov_stream_t stream)
{
    LOCOV_REQUIRE(K >= 0, "locov_nms_sorted: K < 0");
    LOCOV_REQUIRE(num_keep, "locov_nms_sorted: null num_keep");
    hipStream_t s = as_stream(stream);
    if (K == 0) {
        hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int), s);
        if (e != hipSuccess) return set_error(LOCOV_ERR_LAUNCH, "locov_nms_sorted: memset: %s", hipGetErrorString(e));
        return LOCOV_OK;
    }
    LOCOV_REQUIRE(boxes_sorted && workspace && keep, "locov_nms_sorted: null pointer");
    LOCOV_REQUIRE((uintptr_t)boxes_sorted % 16 == 0, "locov_nms_sorted: boxes must be 16-byte aligned");
    LOCOV_REQUIRE(K <= 1 << 20, "locov_nms_sorted: K too large");
    const int cb = (int)((K + 63) / 64);
    LOCOV_REQUIRE((size_t)cb * 8 <= 64 * 1024, "locov_nms_sorted: K too large for the LDS removed-set");
    // rows never write the blocks left of their own: clear the matrix first
    hipError_t e = hipMemsetAsync(workspace, 0, (size_t)locov_nms_workspace_bytes(K), s);
    if (e != hipSuccess) return set_error(LOCOV_ERR_LAUNCH, "locov_nms_sorted: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb), dim3(64), 0, s, reinterpret_cast<const float4 *>(boxes_sorted), (int)K,
                       iou_threshold, reinterpret_cast<unsigned long long *>(workspace), cb);
    int rc = check_launch("locov_nms_sorted(mask)");
    if (rc) return rc;
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(256), (size_t)cb * 8, s,
                       reinterpret_cast<const unsigned long long *>(workspace), (int)K, cb, keep, num_keep);
    return check_launch("locov_nms_sorted(sweep)");
}

}  // extern "C"
