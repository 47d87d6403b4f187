// ROIAlign forward/backward (NCHW contract path) and FPN level assignment for gfx950.
//
// Replaces [D2-upstream] ROIPooler -> ROIAlign -> torchvision.ops.roi_align reached from
// ovr/modeling/roi_heads/roi_emb_heads.py:243-245, and assign_boxes_to_levels.
//
// Design (MI355X): one workgroup = one ROI x one channel tile.  Bilinear sampling is
// separable, so the workgroup first builds two tiny per-axis tables in LDS
// (pooled*grid entries each: low/high index + low/high weight) that every channel of the
// tile reuses; the main loop is then 4 gathers + weights per sample with no coordinate
// math.  For a fixed ROI the output block out[r, c0:c0+CT, :, :] is one contiguous run of
// CT*ph*pw floats, so lanes write consecutive addresses (the kernel is HBM-write-bound:
// 0.80 MB per proposal at C=1024, P=14).  Coordinates and weights use un-fused fp32
// (FMA contraction off), which makes the result bit-identical to the oracle.
#include "common.h"

// The oracle defines un-fused fp32 arithmetic; HIP's __f*_rn helpers are plain operators and
// would be contracted to FMA, so contraction is switched off for this whole file.
#pragma clang fp contract(off)

namespace locov {

// plain operators compiled under contract(off): one IEEE rounding each, never fused
__device__ __forceinline__ float f_add(float a, float b) { return a + b; }
__device__ __forceinline__ float f_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float f_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float f_div(float a, float b) { return a / b; }

struct LevelDesc {
    const float *feat;
    int H, W;
    float scale;
};
struct LevelTable {
    LevelDesc lv[LOCOV_MAX_LEVELS];
};

struct AxisSample {
    int lo, hi;   // element offsets (already multiplied by W on the y axis)
    float wl, wh; // weight of the high tap (l = frac) and of the low tap (h = 1 - frac)
};

struct RoiGeom {
    float start_h, start_w, bin_h, bin_w, count;
    int grid_h, grid_w;
};

// One axis sample exactly as torchvision's bilinear pre-calc does it (see oracle_precalc).
__device__ __forceinline__ AxisSample axis_sample(float start, float bin, int p, int i, int grid, int size,
                                                  int stride)
{
    // v = start + p*bin + ((i + .5f) * bin) / grid      -- left-to-right, un-fused
    float v = f_add(f_add(start, f_mul((float)p, bin)),
                        f_div(f_mul(f_add((float)i, .5f), bin), (float)grid));
    AxisSample s;
    if (v < -1.0f || v > (float)size) {  // sample outside the map contributes zero
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
    const float l = f_sub(v, (float)lo);
    s.lo = lo * stride; s.hi = hi * stride;
    s.wl = l; s.wh = f_sub(1.f, l);
    return s;
}

__device__ __forceinline__ RoiGeom roi_geom(const float *roi, float scale, int ph, int pw, int sampling_ratio,
                                            int aligned)
{
    RoiGeom g;
    const float off = aligned ? 0.5f : 0.0f;
    g.start_w = f_sub(f_mul(roi[1], scale), off);
    g.start_h = f_sub(f_mul(roi[2], scale), off);
    const float end_w = f_sub(f_mul(roi[3], scale), off);
    const float end_h = f_sub(f_mul(roi[4], scale), off);
    float rw = f_sub(end_w, g.start_w), rh = f_sub(end_h, g.start_h);
    if (!aligned) {
        rw = fmaxf(rw, 1.f);
        rh = fmaxf(rh, 1.f);
    }
    g.bin_h = f_div(rh, (float)ph);
    g.bin_w = f_div(rw, (float)pw);
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(g.bin_h);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(g.bin_w);
    const int prod = gh * gw;
    g.count = (float)(prod > 1 ? prod : 1);
    g.grid_h = gh > 0 ? gh : 0;
    g.grid_w = gw > 0 ? gw : 0;
    return g;
}

constexpr int kRoiThreads = 256;
constexpr int kMaxAxisEntries = 1024;  // per axis; 2 * 1024 * 16 B = 32 KiB LDS

// FWD = true : out[r,c,ph,pw] = mean over the sample grid of the bilinear taps.
// FWD = false: scatter grad_out[r,c,ph,pw] * w / count into grad_feat with fp32 atomics.
template <bool FWD>
__global__ __launch_bounds__(kRoiThreads) void roi_align_nchw_kernel(
    LevelTable levels, int num_levels, const int64_t *__restrict__ level_of_roi, int N, int C,
    const float *__restrict__ rois, int PH, int PW, int sampling_ratio, int aligned, int c_tile,
    float *__restrict__ out_or_gradfeat, const float *__restrict__ grad_out)
{
    __shared__ AxisSample ytab[kMaxAxisEntries];
    __shared__ AxisSample xtab[kMaxAxisEntries];

    const int64_t r = blockIdx.x;
    const int c0 = blockIdx.y * c_tile;
    const int cn = min(c_tile, C - c0);
    const float *roi = rois + r * 5;
    const int lvl = (num_levels > 1 && level_of_roi) ? (int)level_of_roi[r] : 0;
    const LevelDesc L = levels.lv[(lvl >= 0 && lvl < num_levels) ? lvl : 0];
    const int b = (int)roi[0];
    RoiGeom g = roi_geom(roi, L.scale, PH, PW, sampling_ratio, aligned);
    if (b < 0 || b >= N || lvl < 0 || lvl >= num_levels) {  // bad batch / level index: no samples -> zeros
        g.grid_h = 0;
        g.grid_w = 0;
    }
    const int ny = PH * g.grid_h, nx = PW * g.grid_w;
    const bool use_lds = ny <= kMaxAxisEntries && nx <= kMaxAxisEntries;

    if (use_lds) {
        for (int t = threadIdx.x; t < ny; t += kRoiThreads)
            ytab[t] = axis_sample(g.start_h, g.bin_h, t / g.grid_h, t % g.grid_h, g.grid_h, L.H, L.W);
        for (int t = threadIdx.x; t < nx; t += kRoiThreads)
            xtab[t] = axis_sample(g.start_w, g.bin_w, t / g.grid_w, t % g.grid_w, g.grid_w, L.W, 1);
    }
    __syncthreads();

    const int bins = PH * PW;
    const int total = cn * bins;
    const int64_t plane_sz = (int64_t)L.H * L.W;
    const float *fbase = L.feat + ((int64_t)b * C + c0) * plane_sz;
    const int64_t obase = (r * C + c0) * (int64_t)bins;

    for (int o = threadIdx.x; o < total; o += kRoiThreads) {
        const int c = o / bins;
        const int bin = o - c * bins;
        const int ph = bin / PW;
        const int pw = bin - ph * PW;
        if (FWD) {
            const float *plane = fbase + c * plane_sz;
            float acc = 0.f;
            for (int iy = 0; iy < g.grid_h; iy++) {
                const AxisSample ys = use_lds ? ytab[ph * g.grid_h + iy]
                                              : axis_sample(g.start_h, g.bin_h, ph, iy, g.grid_h, L.H, L.W);
                for (int ix = 0; ix < g.grid_w; ix++) {
                    const AxisSample xs = use_lds ? xtab[pw * g.grid_w + ix]
                                                  : axis_sample(g.start_w, g.bin_w, pw, ix, g.grid_w, L.W, 1);
                    const float v1 = plane[ys.lo + xs.lo], v2 = plane[ys.lo + xs.hi];
                    const float v3 = plane[ys.hi + xs.lo], v4 = plane[ys.hi + xs.hi];
                    const float w1 = f_mul(ys.wh, xs.wh), w2 = f_mul(ys.wh, xs.wl);
                    const float w3 = f_mul(ys.wl, xs.wh), w4 = f_mul(ys.wl, xs.wl);
                    // ((w1*v1 + w2*v2) + w3*v3) + w4*v4, then accumulate: oracle order
                    const float s = f_add(
                        f_add(f_add(f_mul(w1, v1), f_mul(w2, v2)), f_mul(w3, v3)),
                        f_mul(w4, v4));
                    acc = f_add(acc, s);
                }
            }
            out_or_gradfeat[obase + o] = f_div(acc, g.count);
        } else {
            float *plane = out_or_gradfeat + ((int64_t)b * C + c0 + c) * plane_sz;
            const float gv = grad_out[obase + o];
            for (int iy = 0; iy < g.grid_h; iy++) {
                const AxisSample ys = use_lds ? ytab[ph * g.grid_h + iy]
                                              : axis_sample(g.start_h, g.bin_h, ph, iy, g.grid_h, L.H, L.W);
                for (int ix = 0; ix < g.grid_w; ix++) {
                    const AxisSample xs = use_lds ? xtab[pw * g.grid_w + ix]
                                                  : axis_sample(g.start_w, g.bin_w, pw, ix, g.grid_w, L.W, 1);
                    const float w1 = ys.wh * xs.wh, w2 = ys.wh * xs.wl, w3 = ys.wl * xs.wh, w4 = ys.wl * xs.wl;
                    if (w1 != 0.f) atomicAdd(plane + ys.lo + xs.lo, gv * w1 / g.count);
                    if (w2 != 0.f) atomicAdd(plane + ys.lo + xs.hi, gv * w2 / g.count);
                    if (w3 != 0.f) atomicAdd(plane + ys.hi + xs.lo, gv * w3 / g.count);
                    if (w4 != 0.f) atomicAdd(plane + ys.hi + xs.hi, gv * w4 / g.count);
                }
            }
        }
    }
}

// One lane per box.  log2 is evaluated in fp64 and rounded once to fp32 (a correctly
// rounded log2f), the same definition as oracle_level_assign, so the integer result does
// not depend on a device libm's last-ulp behaviour.
__global__ __launch_bounds__(256) void level_assign_kernel(const float *__restrict__ boxes, int64_t R,
                                                           int min_level, int max_level, float canon_size,
                                                           float canon_level, int64_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const float4 b = reinterpret_cast<const float4 *>(boxes)[i];
    const float area = f_mul(f_sub(b.z, b.x), f_sub(b.w, b.y));
    const float size = __fsqrt_rn(area);
    const float arg = f_add(f_div(size, canon_size), 1e-8f);
    const float l2 = (float)log2((double)arg);
    float lvl = floorf(f_add(canon_level, l2));
    if (!(lvl >= (float)min_level)) lvl = (float)min_level;  // also maps NaN to min_level
    if (lvl > (float)max_level) lvl = (float)max_level;
    out[i] = (int64_t)lvl - min_level;
}

static int launch_roi_align(bool fwd, const LevelTable &tab, int num_levels, const int64_t *levels, int N, int C,
                            const float *rois, int64_t R, int PH, int PW, int sampling_ratio, int aligned,
                            float *out, const float *grad_out, hipStream_t s, const char *what)
{
    // channel tile: enough outputs per workgroup to amortise the table build, enough
    // workgroups (>> 256 CUs) to fill the chip.
    int c_tile = 32;
    while (c_tile > 1 && R * ceil_div(C, c_tile) < 2048) c_tile >>= 1;
    dim3 grid((unsigned)R, (unsigned)ceil_div(C, c_tile));
    if (fwd)
        hipLaunchKernelGGL(roi_align_nchw_kernel<true>, grid, dim3(kRoiThreads), 0, s, tab, num_levels, levels,
                           N, C, rois, PH, PW, sampling_ratio, aligned, c_tile, out, nullptr);
    else
        hipLaunchKernelGGL(roi_align_nchw_kernel<false>, grid, dim3(kRoiThreads), 0, s, tab, num_levels, levels,
                           N, C, rois, PH, PW, sampling_ratio, aligned, c_tile, out, grad_out);
    return check_launch(what);
}

static int check_roi_args(const char *what, const void *feat, int N, int C, int H, int W, const float *rois,
                          int64_t R, int ph, int pw, float scale, const void *out)
{
    if (R < 0) return set_error(LOCOV_ERR_INVALID_ARG, "%s: R < 0", what);
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0)
        return set_error(LOCOV_ERR_INVALID_ARG, "%s: bad feature shape [%d,%d,%d,%d]", what, N, C, H, W);
    if (ph <= 0 || pw <= 0) return set_error(LOCOV_ERR_INVALID_ARG, "%s: bad pooled size %dx%d", what, ph, pw);
    if (!(scale > 0.f)) return set_error(LOCOV_ERR_INVALID_ARG, "%s: spatial_scale must be > 0", what);
    if (R > 0 && (!feat || !rois || !out)) return set_error(LOCOV_ERR_INVALID_ARG, "%s: null pointer", what);
    if (R > 0x7fffffffLL) return set_error(LOCOV_ERR_INVALID_ARG, "%s: R too large", what);
    return LOCOV_OK;
}

}  // namespace locov

using namespace locov;

extern "C" {

int locov_level_assign(const float *boxes, int64_t R, int min_level, int max_level, int canonical_box_size,
                       int canonical_level, int64_t *levels, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0, "locov_level_assign: R < 0");
    LOCOV_REQUIRE(min_level <= max_level, "locov_level_assign: min_level > max_level");
    LOCOV_REQUIRE(canonical_box_size > 0, "locov_level_assign: canonical_box_size must be > 0");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(boxes && levels, "locov_level_assign: null pointer");
    LOCOV_REQUIRE((uintptr_t)boxes % 16 == 0, "locov_level_assign: boxes must be 16-byte aligned");
    hipLaunchKernelGGL(level_assign_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, as_stream(stream),
                       boxes, R, min_level, max_level, (float)canonical_box_size, (float)canonical_level,
                       levels);
    return check_launch("locov_level_assign");
}

int locov_roi_align_fwd(const float *feat, int N, int C, int H, int W, const float *rois, int64_t R,
                        int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                        float *out, locov_stream_t stream)
{
    int rc = check_roi_args("locov_roi_align_fwd", feat, N, C, H, W, rois, R, pooled_h, pooled_w, spatial_scale,
                            out);
    if (rc != LOCOV_OK || R == 0) return rc;
    LevelTable tab = {};
    tab.lv[0] = {feat, H, W, spatial_scale};
    return launch_roi_align(true, tab, 1, nullptr, N, C, rois, R, pooled_h, pooled_w, sampling_ratio, aligned, out,
                            nullptr, as_stream(stream), "locov_roi_align_fwd");
}

int locov_roi_align_bwd(const float *grad_out, int N, int C, int H, int W, const float *rois, int64_t R,
                        int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                        float *grad_feat, locov_stream_t stream)
{
    int rc = check_roi_args("locov_roi_align_bwd", grad_out, N, C, H, W, rois, R, pooled_h, pooled_w,
                            spatial_scale, grad_feat);
    if (rc != LOCOV_OK || R == 0) return rc;
    LevelTable tab = {};
    tab.lv[0] = {grad_feat, H, W, spatial_scale};  // feat pointer unused in bwd; H/W/scale are
    return launch_roi_align(false, tab, 1, nullptr, N, C, rois, R, pooled_h, pooled_w, sampling_ratio, aligned,
                            grad_feat, grad_out, as_stream(stream), "locov_roi_align_bwd");
}

int locov_roi_align_levels_fwd(const float *const *feats_host, const int *H_host, const int *W_host,
                               const float *scales_host, int num_levels, int N, int C, const float *rois,
                               const int64_t *levels, int64_t R, int pooled_h, int pooled_w, int sampling_ratio,
                               int aligned, float *out, locov_stream_t stream)
{
    LOCOV_REQUIRE(num_levels >= 1 && num_levels <= LOCOV_MAX_LEVELS,
                  "locov_roi_align_levels_fwd: num_levels must be in [1,%d]", LOCOV_MAX_LEVELS);
    LOCOV_REQUIRE(feats_host && H_host && W_host && scales_host, "locov_roi_align_levels_fwd: null host array");
    LOCOV_REQUIRE(num_levels == 1 || levels || R == 0, "locov_roi_align_levels_fwd: levels required");
    LevelTable tab = {};
    for (int i = 0; i < num_levels; i++) {
        int rc = check_roi_args("locov_roi_align_levels_fwd", feats_host[i], N, C, H_host[i], W_host[i], rois, R,
                                pooled_h, pooled_w, scales_host[i], out);
        if (rc != LOCOV_OK) return rc;
        tab.lv[i] = {feats_host[i], H_host[i], W_host[i], scales_host[i]};
    }
    if (R == 0) return LOCOV_OK;
    return launch_roi_align(true, tab, num_levels, levels, N, C, rois, R, pooled_h, pooled_w, sampling_ratio,
                            aligned, out, nullptr, as_stream(stream), "locov_roi_align_levels_fwd");
}

}  // extern "C"
