// Backward of the box predictor's dense part (box_emb_head.py:196-211 under autograd), the two entry points SURVEY 8b names
// next to their forwards:
//   locov_pool_fc_bwd   bbox_pred / emb_pred:  grad_x = g_emb W_emb + g_box W_box,  grad_W = g^T x,  grad_b = column sums of g
//   locov_sim_gemm_bwd  cls_score (the similarity GEMM): grad_emb = g_logits . bank,  grad_bank = g_logits^T emb
// Compositions of the library's own GEMMs: data gradients are NT GEMMs against a transposed copy of the (small) weight
// kept in the caller's workspace -- the second one adds onto the first through the epilogue's residual --, weight gradients
// are TN GEMMs (contraction over the rows, no activation-sized transpose), bias gradients a deterministic column sum.
#include "gemm_nt.h"

namespace locov {

namespace {

// out[n] = sum_m g[m, n]: one workgroup per 64 columns, 4 row lanes x 64 columns, fixed summation order
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ g, int64_t M, int N, int64_t ld, float *__restrict__ out)
{
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6, n = blockIdx.x * 64 + c;
    float s = 0.f;
    if (n < N)
        for (int64_t m = q; m < M; m += 4) s += g[m * ld + n];
    part[q][c] = s;
    __syncthreads();
    if (q == 0 && n < N) out[n] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

// out[k, n] = w[n, k]  (w [N, K] row-major -> [K, ldo], columns N..ldo-1 zero; ldo - N < 4)
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ w, int N, int K, float *__restrict__ out, int ldo)
{
    __shared__ float t[64][65];
    const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 64, c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    for (int r = r0; r < 64; r += 4)
        if (n0 + r < N && k0 + c < K) t[r][c] = w[(int64_t)(n0 + r) * K + k0 + c];
    __syncthreads();
    for (int r = r0; r < 64; r += 4)
        if (k0 + r < K && n0 + c < ldo) out[(int64_t)(k0 + r) * ldo + n0 + c] = n0 + c < N ? t[c][r] : 0.f;
}

int transpose(const float *w, int N, int K, float *out, int ldo, hipStream_t s)
{
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)ceil_div(K, 64), (unsigned)ceil_div(N, 64)), dim3(256), 0, s, w, N, K, out, ldo);
    return check_launch("transpose");
}

// out[m, 0..Kp) = g[m, 0..K) followed by zeros (the NT GEMM stages 16-byte chunks: K % 4 != 0 operands get a padded copy)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float *__restrict__ g, int64_t M, int K, int Kp, float *__restrict__ out)
{
    const int64_t total = M * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / Kp;
        const int k = (int)(i - m * Kp);
        out[i] = k < K ? g[m * K + k] : 0.f;
    }
}

int64_t align256(int64_t b) { return (b + 255) & ~(int64_t)255; }

}  // namespace

}  // namespace locov

using namespace locov;

extern "C" {

int64_t locov_pool_fc_bwd_workspace_bytes(int64_t R, int C5, int D)
{
    if (R < 0 || C5 <= 0 || D <= 0) return -1;
    const int64_t tn = gemm_tn_workspace_bytes(R, D, C5, 1), tn4 = gemm_tn_workspace_bytes(R, 4, C5, 1);
    return align256((int64_t)D * C5 * 4) + align256((int64_t)4 * C5 * 4) + align256(tn > tn4 ? tn : tn4);
}

int locov_pool_fc_bwd(const float *x, int64_t R, int C5, const float *emb_w, int D, const float *bbox_w, const float *grad_emb,
                      const float *grad_deltas, float *grad_x, float *grad_emb_w, float *grad_emb_b, float *grad_bbox_w,
                      float *grad_bbox_b, void *workspace, int64_t workspace_bytes, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && C5 > 0 && D > 0, "locov_pool_fc_bwd: bad shape R=%lld C5=%d D=%d", (long long)R, C5, D);
    LOCOV_REQUIRE(C5 % 4 == 0 && D % 4 == 0, "locov_pool_fc_bwd: C5 and D must be multiples of 4");
    LOCOV_REQUIRE(workspace && workspace_bytes >= locov_pool_fc_bwd_workspace_bytes(R, C5, D) && (uintptr_t)workspace % 256 == 0,
                  "locov_pool_fc_bwd: workspace too small or misaligned (locov_pool_fc_bwd_workspace_bytes)");
    hipStream_t s = as_stream(stream);
    char *ws = static_cast<char *>(workspace);
    float *wt_emb = reinterpret_cast<float *>(ws);
    float *wt_box = reinterpret_cast<float *>(ws + align256((int64_t)D * C5 * 4));
    float *tn_ws = reinterpret_cast<float *>(ws + align256((int64_t)D * C5 * 4) + align256((int64_t)4 * C5 * 4));
    const int64_t tn_bytes = workspace_bytes - (reinterpret_cast<char *>(tn_ws) - ws);
    int rc;
    if (R == 0) {                                          // empty batch: zero weight gradients, nothing else to write
        if (grad_emb_w && hipMemsetAsync(grad_emb_w, 0, (size_t)D * C5 * 4, s) != hipSuccess) return set_error(LOCOV_ERR_LAUNCH, "locov_pool_fc_bwd: memset");
        if (grad_emb_b && hipMemsetAsync(grad_emb_b, 0, (size_t)D * 4, s) != hipSuccess) return set_error(LOCOV_ERR_LAUNCH, "locov_pool_fc_bwd: memset");
        if (grad_bbox_w && hipMemsetAsync(grad_bbox_w, 0, (size_t)4 * C5 * 4, s) != hipSuccess) return set_error(LOCOV_ERR_LAUNCH, "locov_pool_fc_bwd: memset");
        if (grad_bbox_b && hipMemsetAsync(grad_bbox_b, 0, 16, s) != hipSuccess) return set_error(LOCOV_ERR_LAUNCH, "locov_pool_fc_bwd: memset");
        return LOCOV_OK;
    }
    LOCOV_REQUIRE(grad_emb || grad_deltas, "locov_pool_fc_bwd: no incoming gradient");
    LOCOV_REQUIRE(!grad_x || ((!grad_emb || emb_w) && (!grad_deltas || bbox_w)), "locov_pool_fc_bwd: grad_x needs the weights");
    LOCOV_REQUIRE(!(grad_emb_w || grad_bbox_w) || x, "locov_pool_fc_bwd: weight gradients need x");
    LOCOV_REQUIRE((!grad_emb_w && !grad_emb_b) || grad_emb, "locov_pool_fc_bwd: emb_pred gradients need grad_emb");
    LOCOV_REQUIRE((!grad_bbox_w && !grad_bbox_b) || grad_deltas, "locov_pool_fc_bwd: bbox_pred gradients need grad_deltas");
    if (grad_x) {                                          // [R,D] . ([C5,D])^T  (+ [R,4] . ([C5,4])^T through the residual)
        bool have = false;
        if (grad_emb) {
            if ((rc = transpose(emb_w, D, C5, wt_emb, D, s))) return rc;
            Epilogue e{nullptr, nullptr, nullptr, 0u};
            if ((rc = launch_gemm_nt<float, float>(grad_emb, D, wt_emb, D, grad_x, C5, R, C5, D, e, s, "locov_pool_fc_bwd(grad_x, emb_pred)"))) return rc;
            have = true;
        }
        if (grad_deltas) {
            if ((rc = transpose(bbox_w, 4, C5, wt_box, 4, s))) return rc;
            Epilogue e{nullptr, nullptr, have ? grad_x : nullptr, 0u};
            if ((rc = launch_gemm_nt<float, float>(grad_deltas, 4, wt_box, 4, grad_x, C5, R, C5, 4, e, s, "locov_pool_fc_bwd(grad_x, bbox_pred)"))) return rc;
        }
    }
    if (grad_emb_w && (rc = launch_gemm_tn(grad_emb, D, 0, x, C5, 0, grad_emb_w, C5, 0, R, D, C5, 1, nullptr, tn_ws, tn_bytes, s, "locov_pool_fc_bwd(grad emb_pred.weight)"))) return rc;
    if (grad_bbox_w && (rc = launch_gemm_tn(grad_deltas, 4, 0, x, C5, 0, grad_bbox_w, C5, 0, R, 4, C5, 1, nullptr, tn_ws, tn_bytes, s, "locov_pool_fc_bwd(grad bbox_pred.weight)"))) return rc;
    if (grad_emb_b) {
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ceil_div(D, 64)), dim3(256), 0, s, grad_emb, R, D, (int64_t)D, grad_emb_b);
        if ((rc = check_launch("locov_pool_fc_bwd(grad emb_pred.bias)"))) return rc;
    }
    if (grad_bbox_b) {
        hipLaunchKernelGGL(colsum_kernel, dim3(1), dim3(256), 0, s, grad_deltas, R, 4, (int64_t)4, grad_bbox_b);
        if ((rc = check_launch("locov_pool_fc_bwd(grad bbox_pred.bias)"))) return rc;
    }
    return LOCOV_OK;
}

int64_t locov_sim_gemm_bwd_workspace_bytes(int64_t R, int D, int K1)
{
    if (R < 0 || D <= 0 || K1 <= 0) return -1;
    const int K1p = (K1 + 3) & ~3;
    return align256((int64_t)D * K1p * 4) + align256(K1 % 4 == 0 ? gemm_tn_workspace_bytes(R, K1, D, 1) : R * (int64_t)K1p * 4);
}

int locov_sim_gemm_bwd(const float *grad_logits, const float *emb, const float *bank, int64_t R, int D, int K1, float *grad_emb,
                       float *grad_bank, void *workspace, int64_t workspace_bytes, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && D > 0 && K1 > 0, "locov_sim_gemm_bwd: bad shape R=%lld D=%d K1=%d", (long long)R, D, K1);
    LOCOV_REQUIRE(D % 4 == 0, "locov_sim_gemm_bwd: D must be a multiple of 4");
    LOCOV_REQUIRE(grad_logits || R == 0, "locov_sim_gemm_bwd: null gradient");
    LOCOV_REQUIRE(!grad_emb || bank, "locov_sim_gemm_bwd: grad_emb needs the bank");
    LOCOV_REQUIRE(!grad_bank || emb || R == 0, "locov_sim_gemm_bwd: grad_bank needs the embeddings");
    LOCOV_REQUIRE(workspace && workspace_bytes >= locov_sim_gemm_bwd_workspace_bytes(R, D, K1) && (uintptr_t)workspace % 256 == 0,
                  "locov_sim_gemm_bwd: workspace too small or misaligned (locov_sim_gemm_bwd_workspace_bytes)");
    if (grad_bank && K1 % 4 != 0)
        return set_error(LOCOV_ERR_UNSUPPORTED, "locov_sim_gemm_bwd: grad_bank needs K1 %% 4 == 0 (got %d); the reference freezes the bank "
                                                "(box_emb_head.py:234-235), so it is only ever asked for by tests", K1);
    hipStream_t s = as_stream(stream);
    char *ws = static_cast<char *>(workspace);
    int rc;
    if (R == 0) {
        if (grad_bank && hipMemsetAsync(grad_bank, 0, (size_t)K1 * D * 4, s) != hipSuccess) return set_error(LOCOV_ERR_LAUNCH, "locov_sim_gemm_bwd: memset");
        return LOCOV_OK;
    }
    const int K1p = (K1 + 3) & ~3;
    float *second = reinterpret_cast<float *>(ws + align256((int64_t)D * K1p * 4));
    if (grad_emb) {                                        // [R,K1] . ([D,K1])^T, the contraction zero-padded to K1p
        float *bt = reinterpret_cast<float *>(ws);
        if ((rc = transpose(bank, K1, D, bt, K1p, s))) return rc;
        const float *g = grad_logits;
        if (K1p != K1) {
            const int64_t total = R * (int64_t)K1p;
            hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)(ceil_div(total, 256) < 4096 ? ceil_div(total, 256) : 4096)), dim3(256), 0, s,
                               grad_logits, R, K1, K1p, second);
            if ((rc = check_launch("locov_sim_gemm_bwd(pad)"))) return rc;
            g = second;
        }
        Epilogue e{nullptr, nullptr, nullptr, 0u};
        if ((rc = launch_gemm_nt<float, float>(g, K1p, bt, K1p, grad_emb, D, R, D, K1p, e, s, "locov_sim_gemm_bwd(grad_emb)"))) return rc;
    }
    if (grad_bank) {
        float *tn_ws = second;
        if ((rc = launch_gemm_tn(grad_logits, K1, 0, emb, D, 0, grad_bank, D, 0, R, K1, D, 1, nullptr, tn_ws,
                                 workspace_bytes - align256((int64_t)D * K1p * 4), s, "locov_sim_gemm_bwd(grad_bank)")))
            return rc;
    }
    return LOCOV_OK;
}

}  // extern "C"
