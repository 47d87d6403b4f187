// Channels-last ROIAlign for gfx950: channels on the lanes.
//
// Same arithmetic as roi_align.hip (torchvision roi_align semantics, reached from
// ovr/modeling/roi_heads/roi_emb_heads.py:243-245) but on an [N,H,W,C] feature map: the four
// bilinear taps of a sample are then four contiguous C-vectors, so every wave-level load is a
// fully coalesced run of 16 B (fp32) / 8 B (bf16) per lane and the interpolation weights are
// wave-uniform LDS broadcasts.  The output [R, oh, ow, C] is row-major "pixels x channels" --
// directly the A operand of Res5's first 1x1 convolutions as an NT GEMM.
//
// bin_stride = 2 evaluates only even bins (ph, pw even): with STRIDE_IN_1X1=True both stride-2
// 1x1 convs of Res5 block 0 (roi_emb_heads.py:217-241) read exactly those positions of the
// 14x14 tile, so 3/4 of the pooler's output bytes are never produced.
#include "common.h"

namespace locov {

struct AxisSampleN {
    int lo, hi;   // pixel index along the axis
    float wl, wh;
};

__device__ __forceinline__ AxisSampleN axis_sample_n(float start, float bin, int p, int i, int grid, int size)
{
    float v = __fadd_rn(__fadd_rn(start, __fmul_rn((float)p, bin)),
                        __fdiv_rn(__fmul_rn(__fadd_rn((float)i, .5f), bin), (float)grid));
    AxisSampleN s;
    if (v < -1.0f || v > (float)size) {
        s.lo = 0; s.hi = 0; s.wl = 0.f; s.wh = 0.f;
        return s;
    }
    if (v <= 0.f) v = 0.f;
    int lo = (int)v, hi;
    if (lo >= size - 1) {
        hi = lo = size - 1;
        v = (float)lo;
    } else {
        hi = lo + 1;
    }
    const float l = __fsub_rn(v, (float)lo);
    s.lo = lo; s.hi = hi; s.wl = l; s.wh = __fsub_rn(1.f, l);
    return s;
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 load4(const __bf16 *p)
{
    const bf16x4 v = *reinterpret_cast<const bf16x4 *>(p);
    return float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void store4(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void store4(__bf16 *p, const float4 &v)
{
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4 *>(p) = o;
}

constexpr int kNhwcThreads = 256;
constexpr int kMaxAxisN = 1024;

// grid = (R, OH): one workgroup = one ROI x one output row of bins.
template <typename TIn, typename TOut>
__global__ __launch_bounds__(kNhwcThreads) void roi_align_nhwc_kernel(
    const TIn *__restrict__ feat, int N, int H, int W, int C, const float *__restrict__ rois, int PH, int PW,
    float scale, int sampling_ratio, int aligned, int bin_stride, int OH, int OW, int pos_major,
    TOut *__restrict__ out)
{
    __shared__ AxisSampleN ytab[kMaxAxisN];
    __shared__ AxisSampleN xtab[kMaxAxisN];

    const int64_t r = blockIdx.x;
    const int oh = blockIdx.y;
    const int ph = oh * bin_stride;
    const float *roi = rois + r * 5;
    const int b = (int)roi[0];

    const float off = aligned ? 0.5f : 0.0f;
    const float start_w = __fsub_rn(__fmul_rn(roi[1], scale), off);
    const float start_h = __fsub_rn(__fmul_rn(roi[2], scale), off);
    const float end_w = __fsub_rn(__fmul_rn(roi[3], scale), off);
    const float end_h = __fsub_rn(__fmul_rn(roi[4], scale), off);
    float rw = __fsub_rn(end_w, start_w), rh = __fsub_rn(end_h, start_h);
    if (!aligned) {
        rw = fmaxf(rw, 1.f);
        rh = fmaxf(rh, 1.f);
    }
    const float bin_h = __fdiv_rn(rh, (float)PH), bin_w = __fdiv_rn(rw, (float)PW);
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(bin_h);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(bin_w);
    const int prod = gh * gw;
    const float inv_count = 1.f / (float)(prod > 1 ? prod : 1);
    gh = gh > 0 ? gh : 0;
    gw = gw > 0 ? gw : 0;
    // table of this bin row's y samples and of every (strided) column's x samples
    const int nx = OW * gw;
    const bool use_lds = gh <= kMaxAxisN && nx <= kMaxAxisN;
    if (use_lds) {
        for (int t = threadIdx.x; t < gh; t += kNhwcThreads) ytab[t] = axis_sample_n(start_h, bin_h, ph, t, gh, H);
        for (int t = threadIdx.x; t < nx; t += kNhwcThreads)
            xtab[t] = axis_sample_n(start_w, bin_w, (t / gw) * bin_stride, t % gw, gw, W);
    }
    __syncthreads();

    const int c4n = C >> 2;
    const int total = OW * c4n;
    const bool valid_b = b >= 0 && b < N;
    const TIn *img = feat + (int64_t)(valid_b ? b : 0) * H * W * C;
    // ROI-major: out[r][oh][ow][c]; position-major: out[oh][ow][r][c] (R = gridDim.x rows per position)
    const int64_t ow_stride = pos_major ? (int64_t)gridDim.x * C : (int64_t)C;
    TOut *orow = pos_major ? out + ((int64_t)oh * OW * gridDim.x + r) * C : out + ((r * OH + oh) * (int64_t)OW) * C;
    for (int o = threadIdx.x; o < total; o += kNhwcThreads) {
        const int ow = o / c4n;
        const int c = (o - ow * c4n) << 2;
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        if (valid_b) {
            for (int iy = 0; iy < gh; iy++) {
                const AxisSampleN ys = use_lds ? ytab[iy] : axis_sample_n(start_h, bin_h, ph, iy, gh, H);
                const TIn *row_lo = img + (int64_t)ys.lo * W * C + c;
                const TIn *row_hi = img + (int64_t)ys.hi * W * C + c;
                for (int ix = 0; ix < gw; ix++) {
                    const AxisSampleN xs = use_lds ? xtab[ow * gw + ix]
                                                   : axis_sample_n(start_w, bin_w, ow * bin_stride, ix, gw, W);
                    const float w1 = ys.wh * xs.wh, w2 = ys.wh * xs.wl, w3 = ys.wl * xs.wh, w4 = ys.wl * xs.wl;
                    const float4 v1 = load4(row_lo + (int64_t)xs.lo * C), v2 = load4(row_lo + (int64_t)xs.hi * C);
                    const float4 v3 = load4(row_hi + (int64_t)xs.lo * C), v4 = load4(row_hi + (int64_t)xs.hi * C);
                    // this file is built with -ffp-contract=off (exact coordinates); the
                    // accumulation asks for FMA explicitly
                    acc.x = fmaf(w4, v4.x, fmaf(w3, v3.x, fmaf(w2, v2.x, fmaf(w1, v1.x, acc.x))));
                    acc.y = fmaf(w4, v4.y, fmaf(w3, v3.y, fmaf(w2, v2.y, fmaf(w1, v1.y, acc.y))));
                    acc.z = fmaf(w4, v4.z, fmaf(w3, v3.z, fmaf(w2, v2.z, fmaf(w1, v1.z, acc.z))));
                    acc.w = fmaf(w4, v4.w, fmaf(w3, v3.w, fmaf(w2, v2.w, fmaf(w1, v1.w, acc.w))));
                }
            }
        }
        acc.x *= inv_count; acc.y *= inv_count; acc.z *= inv_count; acc.w *= inv_count;
        store4(orow + ow * ow_stride + c, acc);
    }
}

// [N,C,H,W] f32 -> [N,H,W,C] (f32 / bf16): 64x64 LDS-tiled transpose of the [C, HW] matrix of
// each image (coalesced on both sides; +1 padding keeps the column reads conflict-free).
template <typename TOut>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *__restrict__ in, int C, int HW,
                                                           TOut *__restrict__ out)
{
    __shared__ float tile[64][65];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const float *src = in + (int64_t)n * C * HW;
    TOut *dst = out + (int64_t)n * C * HW;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, p = p0 + tx;
        tile[i][tx] = (c < C && p < HW) ? src[(int64_t)c * HW + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int p = p0 + i, c = c0 + tx;
        if (p < HW && c < C) dst[(int64_t)p * C + c] = (TOut)tile[tx][i];
    }
}

}  // namespace locov

using namespace locov;

extern "C" {

int locov_nchw_to_nhwc(const float *in, int N, int C, int H, int W, void *out, int out_dtype, locov_stream_t stream)
{
    LOCOV_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "locov_nchw_to_nhwc: bad shape");
    LOCOV_REQUIRE(in && out, "locov_nchw_to_nhwc: null pointer");
    LOCOV_REQUIRE(out_dtype == LOCOV_F32 || out_dtype == LOCOV_BF16, "locov_nchw_to_nhwc: bad out_dtype %d", out_dtype);
    const int HW = H * W;
    dim3 grid((unsigned)ceil_div(HW, 64), (unsigned)ceil_div(C, 64), (unsigned)N);
    if (out_dtype == LOCOV_F32)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, as_stream(stream), in, C, HW, (float *)out);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, grid, dim3(256), 0, as_stream(stream), in, C, HW,
                           (__bf16 *)out);
    return check_launch("locov_nchw_to_nhwc");
}

int locov_roi_align_nhwc_fwd(const void *feat, int feat_dtype, int N, int H, int W, int C, const float *rois,
                             int64_t R, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                             int aligned, int bin_stride, int pos_major, void *out, int out_dtype, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0, "locov_roi_align_nhwc_fwd: R < 0");
    LOCOV_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "locov_roi_align_nhwc_fwd: bad feature shape");
    LOCOV_REQUIRE(pooled_h > 0 && pooled_w > 0, "locov_roi_align_nhwc_fwd: bad pooled size");
    LOCOV_REQUIRE(spatial_scale > 0.f, "locov_roi_align_nhwc_fwd: spatial_scale must be > 0");
    LOCOV_REQUIRE(bin_stride == 1 || bin_stride == 2, "locov_roi_align_nhwc_fwd: bin_stride must be 1 or 2");
    LOCOV_REQUIRE(C % 4 == 0, "locov_roi_align_nhwc_fwd: C must be a multiple of 4");
    LOCOV_REQUIRE((feat_dtype == LOCOV_F32 || feat_dtype == LOCOV_BF16) && (out_dtype == LOCOV_F32 || out_dtype == LOCOV_BF16),
                  "locov_roi_align_nhwc_fwd: bad dtype");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(feat && rois && out, "locov_roi_align_nhwc_fwd: null pointer");
    LOCOV_REQUIRE(R <= 0x7fffffffLL, "locov_roi_align_nhwc_fwd: R too large");
    const int OH = (pooled_h + bin_stride - 1) / bin_stride, OW = (pooled_w + bin_stride - 1) / bin_stride;
    dim3 grid((unsigned)R, (unsigned)OH);
    hipStream_t s = as_stream(stream);
#define LOCOV_LAUNCH_NHWC(TI, TO)                                                                                   \
    hipLaunchKernelGGL((roi_align_nhwc_kernel<TI, TO>), grid, dim3(kNhwcThreads), 0, s, (const TI *)feat, N, H, W, C, \
                       rois, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, bin_stride, OH, OW, pos_major, (TO *)out)
    if (feat_dtype == LOCOV_F32 && out_dtype == LOCOV_F32) LOCOV_LAUNCH_NHWC(float, float);
    else if (feat_dtype == LOCOV_F32 && out_dtype == LOCOV_BF16) LOCOV_LAUNCH_NHWC(float, __bf16);
    else if (feat_dtype == LOCOV_BF16 && out_dtype == LOCOV_F32) LOCOV_LAUNCH_NHWC(__bf16, float);
    else LOCOV_LAUNCH_NHWC(__bf16, __bf16);
#undef LOCOV_LAUNCH_NHWC
    return check_launch("locov_roi_align_nhwc_fwd");
}

}  // extern "C"
