// Spatial mean, row normalisation and the fused box-head entry point for gfx950.
//
// Replaces box_features.mean(dim=[2,3]) (roi_emb_heads.py:262,344,356), normalize_vec /
// standardize_vec (logged_module.py:55-72) and EmbeddingFastRCNNOutputLayers.forward
// (box_emb_head.py:179-212).
#include "gemm_nt.h"

namespace locov {

constexpr int kMeanRows = 256;

// x [rows, HW] -> out [rows] (rows = R*C).  HBM-read-bound (0.40 MB per proposal at
// C5=2048, HW=49).  A workgroup streams 256 consecutive rows (one contiguous run of
// 256*HW floats) into LDS with 16-byte coalesced loads; each lane then sums its own row
// from LDS in index order (row stride HW floats: conflict-free for odd HW), which is the
// sequential fp32 sum the oracle defines.
__global__ __launch_bounds__(kMeanRows) void spatial_mean_rows_kernel(const float *__restrict__ x, int64_t rows,
                                                                      int HW, float inv_hw_den,
                                                                      float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int64_t row0 = (int64_t)blockIdx.x * kMeanRows;
    const int nrows = (int)min((int64_t)kMeanRows, rows - row0);
    const int64_t base = row0 * HW;
    const int n = nrows * HW;
    const float *src = x + base;
    if ((base & 3) == 0) {  // 16-byte aligned run
        const int n4 = n >> 2;
        for (int i = threadIdx.x; i < n4; i += kMeanRows)
            reinterpret_cast<float4 *>(tile)[i] = reinterpret_cast<const float4 *>(src)[i];
        for (int i = (n4 << 2) + threadIdx.x; i < n; i += kMeanRows) tile[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < n; i += kMeanRows) tile[i] = src[i];
    }
    __syncthreads();
    if ((int)threadIdx.x < nrows) {
        const float *p = tile + threadIdx.x * HW;
        float s = 0.f;
        for (int k = 0; k < HW; k++) s = __fadd_rn(s, p[k]);
        out[row0 + threadIdx.x] = __fdiv_rn(s, inv_hw_den);
    }
}

// Fallback for large HW (row does not fit the LDS tile): one wave per row.
__global__ __launch_bounds__(256) void spatial_mean_wave_kernel(const float *__restrict__ x, int64_t rows, int HW,
                                                                float *__restrict__ out)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float *p = x + row * HW;
    float s = 0.f;
    for (int k = lane; k < HW; k += 64) s += p[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[row] = s / (float)HW;
}

// channels-last: x [R, HW, C] -> out [R, C]; lanes across channels (16 B each), loop over HW.
// roi_stride / pos_stride (in float4 units) select the row order: ROI-major [R,HW,C] -> (HW*C/4, C/4),
// position-major [HW,R,C] -> (C/4, R*C/4).
__global__ __launch_bounds__(256) void spatial_mean_nhwc_kernel(const float *__restrict__ x, int64_t R, int C,
                                                                int HW, int64_t roi_stride, int64_t pos_stride,
                                                                float *__restrict__ out)
{
    const int c4n = C >> 2;
    const int64_t total = R * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c4n;
        const int c4 = (int)(i - r * c4n);
        const float4 *p = reinterpret_cast<const float4 *>(x) + r * roi_stride + c4;
        float4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < HW; k++) {
            const float4 v = p[(int64_t)k * pos_stride];
            s.x = __fadd_rn(s.x, v.x); s.y = __fadd_rn(s.y, v.y);
            s.z = __fadd_rn(s.z, v.z); s.w = __fadd_rn(s.w, v.w);
        }
        const float d = (float)HW;
        float4 o = {__fdiv_rn(s.x, d), __fdiv_rn(s.y, d), __fdiv_rn(s.z, d), __fdiv_rn(s.w, d)};
        reinterpret_cast<float4 *>(out + r * C)[c4] = o;
    }
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// One wave per row, wavefront-shuffle reductions.  L2: y = x / max(||x||_2, eps)
// (F.normalize semantics).  Standardise: y = (x - mean) / (std_unbiased + eps).
__global__ __launch_bounds__(256) void rownorm_kernel(const float *__restrict__ x, int64_t R, int D, int mode,
                                                      float eps, float *__restrict__ y)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    const float *p = x + row * D;
    float *q = y + row * D;
    if (mode == LOCOV_NORM_L2) {
        float ss = 0.f;
        for (int k = lane; k < D; k += 64) ss += p[k] * p[k];
        const float nrm = sqrtf(wave_sum(ss));
        const float den = fmaxf(nrm, eps);
        for (int k = lane; k < D; k += 64) q[k] = p[k] / den;
    } else if (mode == LOCOV_NORM_STANDARDIZE) {
        float s = 0.f;
        for (int k = lane; k < D; k += 64) s += p[k];
        const float mean = wave_sum(s) / (float)D;
        float vs = 0.f;
        for (int k = lane; k < D; k += 64) {
            const float d = p[k] - mean;
            vs += d * d;
        }
        const float var = wave_sum(vs) / (float)(D - 1);
        const float den = sqrtf(var) + eps;
        for (int k = lane; k < D; k += 64) q[k] = (p[k] - mean) / den;
    } else {
        for (int k = lane; k < D; k += 64) q[k] = p[k];
    }
}

// Backward of rownorm_kernel: one wave per row.
//   L2          y = x / den, den = max(|x|, eps):   dx = (g - y (y.g)) / den   (den = eps: the clamp is active, dx = g / eps)
//   standardise y = (x - m) / den, den = std + eps (unbiased std):
//               dx = (g - mean(g)) / den - (x - m) (g.(x - m)) / (den^2 std (D - 1))     (std = 0: the second term is 0)
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float *__restrict__ x, const float *__restrict__ g, int64_t R, int D,
                                                          int mode, float eps, float *__restrict__ dx)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    const float *p = x + row * D, *gr = g + row * D;
    float *q = dx + row * D;
    if (mode == LOCOV_NORM_L2) {
        float ss = 0.f, xg = 0.f;
        for (int k = lane; k < D; k += 64) {
            ss += p[k] * p[k];
            xg += p[k] * gr[k];
        }
        const float nrm = sqrtf(wave_sum(ss));
        xg = wave_sum(xg);
        if (nrm > eps) {
            const float inv = 1.f / nrm, c = xg * inv * inv * inv;           // (y.g) y / den = x (x.g) / |x|^3
            for (int k = lane; k < D; k += 64) q[k] = gr[k] * inv - p[k] * c;
        } else {
            for (int k = lane; k < D; k += 64) q[k] = gr[k] / eps;
        }
    } else if (mode == LOCOV_NORM_STANDARDIZE) {
        float s = 0.f, sg = 0.f;
        for (int k = lane; k < D; k += 64) {
            s += p[k];
            sg += gr[k];
        }
        const float mean = wave_sum(s) / (float)D, gmean = wave_sum(sg) / (float)D;
        float vs = 0.f, gd = 0.f;
        for (int k = lane; k < D; k += 64) {
            const float d = p[k] - mean;
            vs += d * d;
            gd += gr[k] * d;
        }
        const float sd = sqrtf(wave_sum(vs) / (float)(D - 1));
        gd = wave_sum(gd);
        const float den = sd + eps;
        const float c = sd > 0.f ? gd / (den * den * sd * (float)(D - 1)) : 0.f;
        for (int k = lane; k < D; k += 64) q[k] = (gr[k] - gmean) / den - (p[k] - mean) * c;
    } else {
        for (int k = lane; k < D; k += 64) q[k] = gr[k];
    }
}

static int spatial_mean(const float *x, int64_t R, int C, int HW, int channels_last, float *out, hipStream_t s)
{
    if (HW == 1) {
        if (out != x) {
            hipError_t e = hipMemcpyAsync(out, x, (size_t)R * C * sizeof(float), hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess)
                return set_error(LOCOV_ERR_LAUNCH, "locov_spatial_mean_fwd: memcpy: %s", hipGetErrorString(e));
        }
        return LOCOV_OK;
    }
    if (channels_last) {
        if (C % 4 != 0) return set_error(LOCOV_ERR_UNSUPPORTED, "locov_spatial_mean_fwd: channels_last needs C %% 4 == 0");
        const int64_t total = R * (C / 4);
        const int64_t want = ceil_div(total, 256);
        const int grid = (int)(want < 256 * 8 ? want : 256 * 8);
        const int64_t c4n = C / 4;
        const int64_t roi_stride = channels_last == 2 ? c4n : (int64_t)HW * c4n;
        const int64_t pos_stride = channels_last == 2 ? R * c4n : c4n;
        hipLaunchKernelGGL(spatial_mean_nhwc_kernel, dim3(grid), dim3(256), 0, s, x, R, C, HW, roi_stride, pos_stride,
                           out);
        return check_launch("locov_spatial_mean_fwd");
    }
    const int64_t rows = R * C;
    const size_t lds = (size_t)kMeanRows * HW * sizeof(float);
    if (lds <= 64 * 1024) {
        hipLaunchKernelGGL(spatial_mean_rows_kernel, dim3((unsigned)ceil_div(rows, kMeanRows)), dim3(kMeanRows), lds, s,
                           x, rows, HW, (float)HW, out);
    } else {
        hipLaunchKernelGGL(spatial_mean_wave_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, s, x, rows, HW,
                           out);
    }
    return check_launch("locov_spatial_mean_fwd");
}

static int rownorm(const float *x, int64_t R, int D, int mode, float eps, float *y, hipStream_t s)
{
    hipLaunchKernelGGL(rownorm_kernel, dim3((unsigned)ceil_div(R, 4)), dim3(256), 0, s, x, R, D, mode, eps, y);
    return check_launch("locov_rownorm_fwd");
}

}  // namespace locov

using namespace locov;

extern "C" {

int locov_spatial_mean_fwd(const float *x, int64_t R, int C, int HW, int channels_last, float *out,
                           locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && C > 0 && HW > 0, "locov_spatial_mean_fwd: bad shape R=%lld C=%d HW=%d", (long long)R, C, HW);
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && out, "locov_spatial_mean_fwd: null pointer");
    LOCOV_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0, "locov_spatial_mean_fwd: misaligned pointer");
    LOCOV_REQUIRE(R * (int64_t)C / 4 < 0x7fffffffLL * 64, "locov_spatial_mean_fwd: too many rows");
    return spatial_mean(x, R, C, HW, channels_last, out, as_stream(stream));
}

int locov_rownorm_fwd(const float *x, int64_t R, int D, int mode, float eps, float *y, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && D > 0, "locov_rownorm_fwd: bad shape R=%lld D=%d", (long long)R, D);
    LOCOV_REQUIRE(mode == LOCOV_NORM_NONE || mode == LOCOV_NORM_L2 || mode == LOCOV_NORM_STANDARDIZE,
                  "locov_rownorm_fwd: unknown mode %d", mode);
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && y, "locov_rownorm_fwd: null pointer");
    return rownorm(x, R, D, mode, eps, y, as_stream(stream));
}

int locov_rownorm_bwd(const float *x, const float *grad_y, int64_t R, int D, int mode, float eps, float *grad_x,
                      locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && D > 0, "locov_rownorm_bwd: bad shape R=%lld D=%d", (long long)R, D);
    LOCOV_REQUIRE(mode == LOCOV_NORM_NONE || mode == LOCOV_NORM_L2 || mode == LOCOV_NORM_STANDARDIZE,
                  "locov_rownorm_bwd: unknown mode %d", mode);
    LOCOV_REQUIRE(mode != LOCOV_NORM_STANDARDIZE || D > 1, "locov_rownorm_bwd: standardise needs D > 1");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && grad_y && grad_x, "locov_rownorm_bwd: null pointer");
    hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((unsigned)ceil_div(R, 4)), dim3(256), 0, as_stream(stream), x, grad_y, R, D, mode, eps,
                       grad_x);
    return check_launch("locov_rownorm_bwd");
}

int locov_box_head_fwd(const float *x, int64_t R, int C5, int HW, int channels_last, const float *emb_w,
                       const float *emb_b, const float *bbox_w, const float *bbox_b, const float *bank,
                       const uint16_t *bank_bf16, int D, int K1, int norm_mode, int sim_dtype, float *pooled,
                       float *deltas, float *emb, uint16_t *emb_bf16, float *logits, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && C5 > 0 && HW > 0 && D > 0 && K1 > 0, "locov_box_head_fwd: bad shape");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && emb_w && bbox_w && pooled && deltas && emb && logits, "locov_box_head_fwd: null pointer");
    LOCOV_REQUIRE(C5 % 4 == 0 && D % 4 == 0, "locov_box_head_fwd: C5 and D must be multiples of 4");
    LOCOV_REQUIRE(sim_dtype == LOCOV_F32 || sim_dtype == LOCOV_BF16, "locov_box_head_fwd: bad sim_dtype %d", sim_dtype);
    if (sim_dtype == LOCOV_BF16)
        LOCOV_REQUIRE(bank_bf16 && emb_bf16 && D % 8 == 0, "locov_box_head_fwd: bf16 similarity needs bank_bf16, emb_bf16, D %% 8 == 0");
    else
        LOCOV_REQUIRE(bank, "locov_box_head_fwd: fp32 similarity needs bank");
    hipStream_t s = as_stream(stream);
    int rc = spatial_mean(x, R, C5, HW, channels_last, pooled, s);                     // roi_emb_heads.py:262
    if (rc) return rc;
    Epilogue e_box{nullptr, bbox_b, nullptr, 0u};                                        // box_emb_head.py:196
    rc = launch_gemm_nt<float, float>(pooled, C5, bbox_w, C5, deltas, 4, R, 4, C5, e_box, s, "locov_box_head_fwd(bbox_pred)");
    if (rc) return rc;
    Epilogue e_emb{nullptr, emb_b, nullptr, 0u};                                         // :206
    rc = launch_gemm_nt<float, float>(pooled, C5, emb_w, C5, emb, D, R, D, C5, e_emb, s, "locov_box_head_fwd(emb_pred)");
    if (rc) return rc;
    if (norm_mode != LOCOV_NORM_NONE) {                                                  // :207-210
        rc = rownorm(emb, R, D, norm_mode, 1e-12f, emb, s);
        if (rc) return rc;
    }
    if (sim_dtype == LOCOV_BF16) {                                                       // :211
        rc = locov_f32_to_bf16(emb, R * (int64_t)D, emb_bf16, stream);
        if (rc) return rc;
        return locov_sim_gemm_bf16(emb_bf16, bank_bf16, R, D, K1, logits, K1, stream);
    }
    Epilogue e_sim{nullptr, nullptr, nullptr, 0u};
    return launch_gemm_nt<float, float>(emb, D, bank, D, logits, K1, R, K1, D, e_sim, s, "locov_box_head_fwd(cls_score)");
}

}  // extern "C"
