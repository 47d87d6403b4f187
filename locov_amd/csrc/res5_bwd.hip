// Small HBM-bound kernels of the Res5 backward pass (the LSM head trains its Res5 convolutions: configs/coco_lsm.yaml:8,
// roi_emb_heads.py:323,343-347 under autograd; FrozenBN only freezes the statistics).  The heavy lifting is done by the
// GEMM kernels (gemm_nt.hip with the mask epilogue for the data gradients, gemm_tn.hip for the weight gradients,
// winograd.hip for the 3x3 convolutions of the 7x7 ROI tiles); here are the operand preparations around them:
//   weight_transpose_scale : Wt[k, n] = s[n] * W[n, k]            (data gradient of a 1x1 convolution as an NT GEMM: dx = g . Wt^T)
//   conv3x3_weight_flip    : w'[c, n, a, b] = s[n] * w[n, c, 2-a, 2-b]   (data gradient of a 3x3 convolution = convolution with w')
//   im2col3x3              : [R*H*W, C] pixel rows -> [R*H*W, 9*C] patches  (weight gradient of a 3x3 convolution on a general grid)
//   conv3x3_wgrad_unpack   : [N, 9*Cin] (k = tap*Cin + c) -> [N, Cin, 3, 3], row-scaled
//   relu_mask / spatial_mean_bwd : ReLU backward of a saved activation, alone or fused with the mean's broadcast
//   rows_subsample / rows_upsample_add : the stride-2 pixel selection of block 0 on the whole grid and its adjoint
#include "gemm_nt.h"

namespace locov {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 64x64 LDS-tiled transpose with a per-source-row scale
__global__ __launch_bounds__(256) void weight_transpose_scale_kernel(const float *__restrict__ w, int N, int K,
                                                                     const float *__restrict__ s, float *__restrict__ out)
{
    __shared__ float tile[64][65];
    const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int n = n0 + i, k = k0 + tx;
        tile[i][tx] = (n < N && k < K) ? w[(int64_t)n * K + k] * (s ? s[n] : 1.f) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int k = k0 + i, n = n0 + tx;
        if (k < K && n < N) out[(int64_t)k * N + n] = tile[tx][i];
    }
}

__global__ __launch_bounds__(256) void conv3x3_weight_flip_kernel(const float *__restrict__ w, int N, int Cin,
                                                                  const float *__restrict__ s, float *__restrict__ out)
{
    const int64_t total = (int64_t)N * Cin * 9;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes the OUTPUT [Cin, N, 3, 3]
        const int t = (int)(i % 9);
        const int64_t cn = i / 9;
        const int n = (int)(cn % N);
        const int64_t c = cn / N;
        out[i] = w[((int64_t)n * Cin + c) * 9 + (8 - t)] * (s ? s[n] : 1.f);
    }
}

// rows are ROI-major: m = (r*H + y)*W + x
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float *__restrict__ x, int64_t M, int H, int W, int C,
                                                        float *__restrict__ col)
{
    const int c4 = C >> 2;
    const int64_t total = M * 9 * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cq = (int)(i % c4);
        const int64_t mt = i / c4;
        const int t = (int)(mt % 9);
        const int64_t m = mt / 9;
        const int px = (int)(m % W), py = (int)((m / W) % H);
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)(py + dy) < (unsigned)H && (unsigned)(px + dx) < (unsigned)W)
            v = *reinterpret_cast<const f32x4 *>(x + (m + (int64_t)dy * W + dx) * C + cq * 4);
        *reinterpret_cast<f32x4 *>(col + (m * 9 + t) * C + cq * 4) = v;
    }
}

__global__ __launch_bounds__(256) void conv3x3_wgrad_unpack_kernel(const float *__restrict__ dwp, int N, int Cin,
                                                                   const float *__restrict__ s, float *__restrict__ dw)
{
    const int64_t total = (int64_t)N * Cin * 9;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes the OUTPUT [N, Cin, 3, 3]
        const int t = (int)(i % 9);
        const int64_t nc = i / 9;
        const int c = (int)(nc % Cin);
        const int64_t n = nc / Cin;
        dw[i] = dwp[n * 9 * Cin + (int64_t)t * Cin + c] * (s ? s[n] : 1.f);
    }
}

// out = act > 0 ? g : 0   (n4 = quads)
// amax_out (here and in spatial_mean_bwd_kernel): optional operand-scale slot receiving max |out| (gemm_nt.h, amax_fold)
__global__ __launch_bounds__(256) void relu_mask_kernel(const float *__restrict__ g, const float *__restrict__ act, int64_t n4,
                                                        float *__restrict__ out, float *amax_out)
{
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4 *>(act)[i];
        f32x4 v = reinterpret_cast<const f32x4 *>(g)[i];
        v[0] = a[0] > 0.f ? v[0] : 0.f; v[1] = a[1] > 0.f ? v[1] : 0.f;
        v[2] = a[2] > 0.f ? v[2] : 0.f; v[3] = a[3] > 0.f ? v[3] : 0.f;
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v[0])), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        reinterpret_cast<f32x4 *>(out)[i] = v;
    }
    if (amax_out != nullptr) amax_fold(amax_out, m);
}

// out[(r*HW + p), c] = (act[(r*HW + p), c] > 0 ? g[r, c] / HW : 0): the spatial mean's broadcast fused with the ReLU
// backward of the stage output (ROI-major rows)
__global__ __launch_bounds__(256) void spatial_mean_bwd_kernel(const float *__restrict__ g, const float *__restrict__ act, int64_t R,
                                                               int C, int HW, float inv, float *__restrict__ out, float *amax_out)
{
    const int c4 = C >> 2;
    const int64_t total = R * HW * c4;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cq = (int)(i % c4);
        const int64_t r = i / ((int64_t)HW * c4);
        f32x4 v = *reinterpret_cast<const f32x4 *>(g + r * C + cq * 4) * inv;
        if (act) {
            const f32x4 a = reinterpret_cast<const f32x4 *>(act)[i];
            v[0] = a[0] > 0.f ? v[0] : 0.f; v[1] = a[1] > 0.f ? v[1] : 0.f;
            v[2] = a[2] > 0.f ? v[2] : 0.f; v[3] = a[3] > 0.f ? v[3] : 0.f;
        }
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v[0])), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        reinterpret_cast<f32x4 *>(out)[i] = v;
    }
    if (amax_out != nullptr) amax_fold(amax_out, m);
}

// channels-last map [N, H, W, C] <-> rows of its stride-2 pixels [N, OH, OW, C], OH = (H+1)/2, OW = (W+1)/2
// FWD: rows = map[::2, ::2];  !FWD: map (all pixels) = rows at the even pixels, 0 elsewhere
template <bool FWD>
__global__ __launch_bounds__(256) void rows_stride2_kernel(const float *__restrict__ src, int N, int H, int W, int C,
                                                           float *__restrict__ dst)
{
    const int c4 = C >> 2, OH = (H + 1) / 2, OW = (W + 1) / 2;
    const int64_t total = FWD ? (int64_t)N * OH * OW * c4 : (int64_t)N * H * W * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cq = (int)(i % c4);
        int64_t p = i / c4;
        if (FWD) {
            const int ox = (int)(p % OW);
            p /= OW;
            const int oy = (int)(p % OH);
            const int64_t n = p / OH;
            reinterpret_cast<f32x4 *>(dst)[i] =
                *reinterpret_cast<const f32x4 *>(src + (((n * H + 2 * oy) * W) + 2 * ox) * C + cq * 4);
        } else {
            const int x = (int)(p % W);
            p /= W;
            const int y = (int)(p % H);
            const int64_t n = p / H;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (!(x & 1) && !(y & 1)) v = *reinterpret_cast<const f32x4 *>(src + (((n * OH + (y >> 1)) * OW) + (x >> 1)) * C + cq * 4);
            reinterpret_cast<f32x4 *>(dst)[i] = v;
        }
    }
}

// GradScaler-style skip ON THE DEVICE (locov_zero_if_raised): when the range-guard word of a backward pass in split arithmetic is
// set, the gradients that pass wrote may hold inf / NaN -- they are zero-filled before anything downstream (an optimizer, DDP's
// all-reduce) can consume them.  With the word clear every workgroup leaves after one scalar load: nothing is read or written.
struct ZeroList {
    float *p[LOCOV_ZERO_LIST_MAX];
    int64_t n[LOCOV_ZERO_LIST_MAX];
};

__global__ __launch_bounds__(256) void zero_if_raised_kernel(ZeroList list, const unsigned *__restrict__ flag)
{
    if (*flag == 0u) return;
    float *p = list.p[blockIdx.y];
    const int64_t n = list.n[blockIdx.y];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = t0; i < n4; i += stride) reinterpret_cast<float4 *>(p)[i] = float4{0.f, 0.f, 0.f, 0.f};
        for (int64_t i = (n4 << 2) + t0; i < n; i += stride) p[i] = 0.f;
    } else {
        for (int64_t i = t0; i < n; i += stride) p[i] = 0.f;
    }
}


static unsigned grid_for(int64_t total) { return (unsigned)(ceil_div(total, 256) < 16384 ? ceil_div(total, 256) : 16384); }

}  // namespace locov

using namespace locov;

extern "C" {

int locov_weight_transpose_scale(const float *w, int N, int K, const float *row_scale, float *out, locov_stream_t stream)
{
    LOCOV_REQUIRE(N > 0 && K > 0, "locov_weight_transpose_scale: bad shape");
    LOCOV_REQUIRE(w && out, "locov_weight_transpose_scale: null pointer");
    hipLaunchKernelGGL(weight_transpose_scale_kernel, dim3((unsigned)ceil_div(K, 64), (unsigned)ceil_div(N, 64)), dim3(256), 0,
                       as_stream(stream), w, N, K, row_scale, out);
    return check_launch("locov_weight_transpose_scale");
}

int locov_conv3x3_weight_flip(const float *w, int N, int Cin, const float *row_scale, float *out, locov_stream_t stream)
{
    LOCOV_REQUIRE(N > 0 && Cin > 0, "locov_conv3x3_weight_flip: bad shape");
    LOCOV_REQUIRE(w && out, "locov_conv3x3_weight_flip: null pointer");
    hipLaunchKernelGGL(conv3x3_weight_flip_kernel, dim3(grid_for((int64_t)N * Cin * 9)), dim3(256), 0, as_stream(stream), w, N, Cin,
                       row_scale, out);
    return check_launch("locov_conv3x3_weight_flip");
}

int locov_im2col3x3_nhwc(const float *x, int64_t R, int H, int W, int C, float *col, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "locov_im2col3x3_nhwc: bad shape (C must be a multiple of 4)");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && col && ((uintptr_t)x | (uintptr_t)col) % 16 == 0, "locov_im2col3x3_nhwc: null or misaligned pointer");
    const int64_t M = R * H * W;
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(M * 9 * (C / 4))), dim3(256), 0, as_stream(stream), x, M, H, W, C, col);
    return check_launch("locov_im2col3x3_nhwc");
}

int locov_conv3x3_wgrad_unpack(const float *dw_packed, int N, int Cin, const float *row_scale, float *dw, locov_stream_t stream)
{
    LOCOV_REQUIRE(N > 0 && Cin > 0, "locov_conv3x3_wgrad_unpack: bad shape");
    LOCOV_REQUIRE(dw_packed && dw, "locov_conv3x3_wgrad_unpack: null pointer");
    hipLaunchKernelGGL(conv3x3_wgrad_unpack_kernel, dim3(grid_for((int64_t)N * Cin * 9)), dim3(256), 0, as_stream(stream), dw_packed, N,
                       Cin, row_scale, dw);
    return check_launch("locov_conv3x3_wgrad_unpack");
}

int locov_relu_mask(const float *g, const float *act, int64_t n, float *out, float *amax_out, locov_stream_t stream)
{
    LOCOV_REQUIRE(n >= 0 && n % 4 == 0, "locov_relu_mask: n must be a non-negative multiple of 4");
    if (n == 0) return LOCOV_OK;
    LOCOV_REQUIRE(g && act && out && ((uintptr_t)g | (uintptr_t)act | (uintptr_t)out) % 16 == 0, "locov_relu_mask: null or misaligned pointer");
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid_for(n / 4)), dim3(256), 0, as_stream(stream), g, act, n / 4, out, amax_out);
    return check_launch("locov_relu_mask");
}

int locov_spatial_mean_bwd(const float *g, const float *act, int64_t R, int C, int HW, float *out, float *amax_out,
                           locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && C > 0 && HW > 0 && C % 4 == 0, "locov_spatial_mean_bwd: bad shape (C must be a multiple of 4)");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(g && out && ((uintptr_t)g | (uintptr_t)act | (uintptr_t)out) % 16 == 0, "locov_spatial_mean_bwd: null or misaligned pointer");
    hipLaunchKernelGGL(spatial_mean_bwd_kernel, dim3(grid_for(R * HW * (C / 4))), dim3(256), 0, as_stream(stream), g, act, R, C, HW,
                       1.0f / (float)HW, out, amax_out);
    return check_launch("locov_spatial_mean_bwd");
}

int locov_rows_stride2(const float *src, int N, int H, int W, int C, int forward, float *dst, locov_stream_t stream)
{
    LOCOV_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "locov_rows_stride2: bad shape (C must be a multiple of 4)");
    LOCOV_REQUIRE(src && dst && ((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "locov_rows_stride2: null or misaligned pointer");
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    if (forward)
        hipLaunchKernelGGL(rows_stride2_kernel<true>, dim3(grid_for((int64_t)N * OH * OW * (C / 4))), dim3(256), 0, as_stream(stream), src, N,
                           H, W, C, dst);
    else
        hipLaunchKernelGGL(rows_stride2_kernel<false>, dim3(grid_for((int64_t)N * H * W * (C / 4))), dim3(256), 0, as_stream(stream), src, N, H,
                           W, C, dst);
    return check_launch("locov_rows_stride2");
}

int locov_zero_if_raised(float *const *tensors, const int64_t *counts, int n_tensors, const unsigned *flag, locov_stream_t stream)
{
    LOCOV_REQUIRE(n_tensors >= 0 && n_tensors <= LOCOV_ZERO_LIST_MAX, "locov_zero_if_raised: 0..%d tensors per call", LOCOV_ZERO_LIST_MAX);
    if (n_tensors == 0) return LOCOV_OK;
    LOCOV_REQUIRE(tensors && counts && flag, "locov_zero_if_raised: null pointer");
    ZeroList list{};
    int used = 0;
    for (int i = 0; i < n_tensors; i++) {
        LOCOV_REQUIRE(counts[i] >= 0, "locov_zero_if_raised: negative element count");
        if (counts[i] == 0) continue;
        LOCOV_REQUIRE(tensors[i] && (uintptr_t)tensors[i] % 4 == 0, "locov_zero_if_raised: null or misaligned tensor %d", i);
        list.p[used] = tensors[i];
        list.n[used] = counts[i];
        used++;
    }
    if (used == 0) return LOCOV_OK;
    hipLaunchKernelGGL(zero_if_raised_kernel, dim3(64, used), dim3(256), 0, as_stream(stream), list, flag);
    return check_launch("locov_zero_if_raised");
}

}  // extern "C"
