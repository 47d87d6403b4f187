// The pooler contract ([R,C,ph,pw] fp32 out of an NCHW map; roi_emb_heads.py:182-187,243-245) with LDS-STAGED PROPOSAL TILES:
// mode LOCOV_ROIALIGN_FAST of locov_roi_align_from_nhwc_fwd_ex, results within 1e-5 of the exact form (SURVEY.md 8d's gate).
//
// The exact form (roi_align_nhwc.hip) issues one 16-byte lane load per bilinear tap and channel quad, in torchvision's sample
// order; what bounds it is the chain load -> wait -> accumulate, seven times per workgroup for a small box and up to 36 taps
// deep per bin for a large one, at 4 workgroups per CU -- not HBM (2 TB/s of output for a 6 TB/s write roof).  Here:
//
//   plan kernel     one thread per ROI: the SEPARABLE form of every bin row / column -- the pixels its samples touch
//                   {first, count <= 6} and the summed bilinear weight of each pixel, so that a bin is
//                   sum_ky sum_kx Wy[ky] Wx[kx] F[y0+ky][x0+kx]: (gh+1)(gw+1) taps instead of 4 gh gw -- and a partition of
//                   the output bins into REGIONS (rpr bin rows x cpr bin columns) whose pixel rectangle fits the LDS window.
//                   1.3 KB per ROI, computed once instead of once per channel slice (32 workgroups per ROI at 1024 channels).
//   execute kernel  one workgroup per (ROI, 32 channels), as the exact form.  Per region: the rectangle's pixels x 32 channels
//                   are fetched ONCE, one coalesced 128-byte line per pixel, all of a thread's loads in flight together
//                   (one memory latency per region instead of one per tap group), then every tap of the region's bins is
//                   an LDS read.  A small box is a single region (its bins share almost all their pixels: the window is
//                   3-12x smaller than the sum of the taps); a large box takes one or two bin rows per region.
//                   Results leave through the same LDS transpose tile as the exact form, 16 bytes per lane.
//
// ROIs the plan cannot express (more than 5 samples per bin and axis, samples more than a pixel apart under a forced
// sampling_ratio, more than 16 bins per axis) are flagged and take the exact arithmetic inside the same launch.
#include "roi_align_common.h"

namespace locov {

namespace {

constexpr int kTlThreads = 256;
constexpr int kTlCh = 32;            // channels per workgroup: one 128-byte line per pixel
constexpr int kTlQN = kTlCh / 4;     // channel quads = lanes per pixel / per bin
constexpr int kTlWinPix = 192;       // LDS window: 24 KiB (with the 25 KiB tile and the plan: 3 workgroups per CU)
constexpr int kTlBins = 16;          // bins per axis a plan covers
constexpr int kTlTaps = 6;           // pixels per bin and axis (sampling grids up to 5)

struct alignas(16) AxisBin {         // 32 bytes: two 16-byte LDS reads
    float w[kTlTaps];
    int first, count;
};

struct alignas(16) RoiPlan {
    int fast;                        // 0: this ROI takes the exact arithmetic
    int batch;                       // image index, -1 = invalid (output zeros)
    int ty, tx;                      // uniform tap-loop bounds: max pixel count over the bin rows / columns
    int rpr, cpr, nrg, ncg;          // bin rows / columns per region, number of row / column groups
    float inv_count;
    int pad[3];
    int rg[kTlBins][2];              // per row group: {first pixel row, rows}
    int cg[kTlBins][2];              // per column group: {first pixel column, columns}
    AxisBin y[kTlBins], x[kTlBins];
};
static_assert(sizeof(RoiPlan) % 16 == 0, "plans are copied 16 bytes per lane");

// {first, count} of every bin of both axes, per thread of the plan kernel, in LDS ([axis][bin][thread]: conflict-free) -- the
// region search below walks them many times, and a walk over the plan in GLOBAL memory is a chain of dependent loads
struct PlanScratch {
    int first[2][kTlBins][64];
    int count[2][kTlBins][64];
};

__device__ void plan_axis(float start, float bin, int P, int grid, int size, AxisBin *out, PlanScratch &ps, int axis, int &max_count, bool &ok)
{
    const int t = threadIdx.x;
    for (int p = 0; p < P; p++) {
        float w[kTlTaps] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int count = 0, base = 0x7fffffff;
        for (int i = 0; i < grid; i++) {
            const AxisSampleN s = axis_sample_n(start, bin, p, i, grid, size);
            if (s.wl != 0.f || s.wh != 0.f) base = min(base, s.lo);
        }
        if (base != 0x7fffffff) {
            for (int i = 0; i < grid; i++) {
                const AxisSampleN s = axis_sample_n(start, bin, p, i, grid, size);
                if (s.wl == 0.f && s.wh == 0.f) continue;      // outside [-1, size]: adds 0 (and still counts in the mean)
                const int klo = s.lo - base, khi = s.hi - base;
                if (khi >= kTlTaps) {                           // samples more than a pixel apart, or too many of them
                    ok = false;
                    break;
                }
#pragma unroll
                for (int k = 0; k < kTlTaps; k++) {             // (static register indices)
                    if (k == klo) w[k] += s.wh;
                    if (k == khi) w[k] += s.wl;
                }
                count = max(count, khi + 1);
            }
        } else {
            base = 0;
        }
        max_count = max(max_count, count);
        ps.first[axis][p][t] = base;
        ps.count[axis][p][t] = count;
        AxisBin b;
#pragma unroll
        for (int k = 0; k < kTlTaps; k++) b.w[k] = w[k];
        b.first = base;
        b.count = count;
        out[p] = b;
    }
}

// pixel span {first, n} of the bins [b0, b1) of one axis
__device__ void span(const PlanScratch &ps, int axis, int b0, int b1, int &first, int &n)
{
    const int t = threadIdx.x;
    int lo = 0x7fffffff, hi = -1;
    for (int b = b0; b < b1; b++) {
        const int c = ps.count[axis][b][t], f = ps.first[axis][b][t];
        if (c > 0) {
            lo = min(lo, f);
            hi = max(hi, f + c);
        }
    }
    first = hi > lo ? lo : 0;
    n = hi > lo ? hi - lo : 0;
}

__device__ int max_span(const PlanScratch &ps, int axis, int P, int per)
{
    int m = 0;
    for (int b0 = 0; b0 < P; b0 += per) {
        int f, n;
        span(ps, axis, b0, min(b0 + per, P), f, n);
        m = max(m, n);
    }
    return m;
}

}  // namespace

__global__ __launch_bounds__(64) void roi_plan_kernel(const float *__restrict__ rois, int64_t R, int N, int H, int W, int PH, int PW,
                                                      float scale, int sampling_ratio, int aligned, RoiPlan *__restrict__ plans)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;                    // (no barrier in this kernel: the LDS scratch is per thread)
    const float *roi = rois + r * 5;
    RoiPlan &pl = plans[r];
    const int b = (int)roi[0];
    const float off = aligned ? 0.5f : 0.0f;
    const float start_w = __fsub_rn(__fmul_rn(roi[1], scale), off), start_h = __fsub_rn(__fmul_rn(roi[2], scale), off);
    const float end_w = __fsub_rn(__fmul_rn(roi[3], scale), off), end_h = __fsub_rn(__fmul_rn(roi[4], scale), off);
    float rw = __fsub_rn(end_w, start_w), rh = __fsub_rn(end_h, start_h);
    if (!aligned) {
        rw = fmaxf(rw, 1.f);
        rh = fmaxf(rh, 1.f);
    }
    const float bin_h = __fdiv_rn(rh, (float)PH), bin_w = __fdiv_rn(rw, (float)PW);
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(bin_h);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(bin_w);
    const int prod = gh * gw;
    pl.inv_count = 1.f / (float)(prod > 1 ? prod : 1);
    pl.batch = (b >= 0 && b < N) ? b : -1;
    pl.pad[0] = pl.pad[1] = pl.pad[2] = 0;
    __shared__ PlanScratch ps;
    bool ok = PH <= kTlBins && PW <= kTlBins && gh >= 0 && gw >= 0 && gh < kTlTaps && gw < kTlTaps;
    int ty = 0, tx = 0;
    if (ok) {
        plan_axis(start_h, bin_h, PH, gh, H, pl.y, ps, 0, ty, ok);
        if (ok) plan_axis(start_w, bin_w, PW, gw, W, pl.x, ps, 1, tx, ok);
    }
    pl.ty = ty;
    pl.tx = tx;
    // regions: as many bin columns as fit beside ONE bin row, then as many bin rows as still fit
    int cpr = PW, rpr = 1;
    if (ok) {
        const int ys1 = max_span(ps, 0, PH, 1);
        while (cpr > 1 && ys1 * max_span(ps, 1, PW, cpr) > kTlWinPix) cpr = (cpr + 1) / 2;
        const int cs = max_span(ps, 1, PW, cpr);
        if (ys1 * cs > kTlWinPix) ok = false;                            // (a single bin wider than the window)
        if (ok) {
            // (a ROI whose whole footprint fits is the common case: try all rows first, then grow from one)
            if (max_span(ps, 0, PH, PH) * cs <= kTlWinPix)
                rpr = PH;
            else
                while (rpr < PH && max_span(ps, 0, PH, rpr + 1) * cs <= kTlWinPix) rpr++;
        }
    }
    pl.fast = ok ? 1 : 0;
    pl.rpr = rpr;
    pl.cpr = cpr;
    const int nrg = (PH + rpr - 1) / rpr, ncg = (PW + cpr - 1) / cpr;
    pl.nrg = nrg;
    pl.ncg = ncg;
    if (ok) {
        for (int g = 0; g < nrg; g++) {
            int f, n;
            span(ps, 0, g * rpr, min(g * rpr + rpr, PH), f, n);
            pl.rg[g][0] = f;
            pl.rg[g][1] = n;
        }
        for (int g = 0; g < ncg; g++) {
            int f, n;
            span(ps, 1, g * cpr, min(g * cpr + cpr, PW), f, n);
            pl.cg[g][0] = f;
            pl.cg[g][1] = n;
        }
    }
}

// grid (R, C / 32).  LDS: tile [32][ts] | union { window [192][32 floats] , exact form's sample tables } | plan
__global__ __launch_bounds__(kTlThreads) void roi_align_tiles_kernel(const float *__restrict__ feat, int N, int H, int W, int C,
                                                                      const float *__restrict__ rois, const RoiPlan *__restrict__ plans,
                                                                      int PH, int PW, float scale, int sampling_ratio, int aligned,
                                                                      float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bins = PH * PW;
    const int ts = bins | 1;                                  // odd row stride of the transpose tile
    float *tile = smem;                                       // [kTlCh][ts]
    float *after_tile = smem + kTlCh * ts + (4 - (kTlCh * ts) % 4) % 4;
    float4 *win = reinterpret_cast<float4 *>(after_tile);
    RoiPlan *pl = reinterpret_cast<RoiPlan *>(after_tile + kTlWinPix * kTlCh);

    const int64_t r = blockIdx.x;
    const int c0 = blockIdx.y * kTlCh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const float4 *src = reinterpret_cast<const float4 *>(plans + r);
        float4 *dst = reinterpret_cast<float4 *>(pl);
        for (int i = tid; i < (int)(sizeof(RoiPlan) / 16); i += kTlThreads) dst[i] = src[i];
    }
    __syncthreads();

    const unsigned xstride = (unsigned)C * (unsigned)sizeof(float), ystride = (unsigned)W * xstride;
    const int q = lane % kTlQN, sub = lane / kTlQN;
    const int cq = c0 + 4 * q;
    const bool c_ok = cq < C;                                 // C % 4 == 0: a quad is all-in or all-out
    const int batch = pl->batch;
    const float *img = feat + (int64_t)(batch >= 0 ? batch : 0) * H * W * C;
    const __amdgpu_buffer_rsrc_t img_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(img), 0, (unsigned)H * ystride, 0x00020000);
    auto tap = [&](unsigned off) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(img_rsrc, off, 0, 0)); };
    const float inv_pw = 1.0f / (float)PW;

    if (pl->fast && batch >= 0) {
        const int rpr = pl->rpr, cpr = pl->cpr, nrg = pl->nrg, ncg = pl->ncg, ty = pl->ty, tx = pl->tx;
        const float inv_count = pl->inv_count;
        const int qq = tid & (kTlQN - 1), p0 = tid / kTlQN;
        const unsigned qoff = (unsigned)min(c0 + 4 * qq, C - 4) * (unsigned)sizeof(float);
        constexpr int PPT = kTlWinPix / (kTlThreads / kTlQN);          // pixels per thread and region, at most
        for (int rgi = 0; rgi < nrg; rgi++) {
            const int ry0 = pl->rg[rgi][0], rh = pl->rg[rgi][1];
            const int b0 = rgi * rpr, nr = min(rpr, PH - b0);
            for (int cgi = 0; cgi < ncg; cgi++) {
                const int rx0 = pl->cg[cgi][0], rw = pl->cg[cgi][1];
                const int d0 = cgi * cpr, nc = min(cpr, PW - d0);
                const int npx = rh * rw;
                if (rgi + cgi > 0) __syncthreads();                     // the previous region's taps are done with the window
                if (npx > 0) {
                    // the region's pixel rectangle -> LDS: pixel p = (row p / rw, column p % rw), its 128 bytes by 8 lanes
                    const float inv_rw = 1.0f / (float)rw;
                    float4 v[PPT];
#pragma unroll
                    for (int i = 0; i < PPT; i++) {
                        const int p = p0 + i * (kTlThreads / kTlQN);
                        if (p < npx) {
                            const int wr = (int)(((float)p + 0.5f) * inv_rw), wc = p - wr * rw;     // exact for these small integers
                            v[i] = tap((unsigned)(ry0 + wr) * ystride + (unsigned)(rx0 + wc) * xstride + qoff);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < PPT; i++) {
                        const int p = p0 + i * (kTlThreads / kTlQN);
                        if (p < npx) win[p * kTlQN + qq] = v[i];
                    }
                }
                __syncthreads();
                // the region's bins: nr x nc of them, 8 lanes (channel quads) each
                const int nb = nr * nc;
                const float inv_nc = 1.0f / (float)nc;
                for (int i0 = 0; i0 < nb; i0 += kTlThreads / kTlQN) {
                    const int i = i0 + wave * (64 / kTlQN) + sub;
                    if (i >= nb || !c_ok) continue;
                    const int br = (int)(((float)i + 0.5f) * inv_nc), bc = i - br * nc;
                    const int ph = b0 + br, pw = d0 + bc;
                    float4 acc = {0.f, 0.f, 0.f, 0.f};
                    if (npx > 0) {
                        const AxisBin *ya = &pl->y[ph], *xa = &pl->x[pw];
                        // both axes' weights in four LDS reads (the tap loops index them statically)
                        const float4 y03 = reinterpret_cast<const float4 *>(ya)[0], y47 = reinterpret_cast<const float4 *>(ya)[1];
                        const float4 x03 = reinterpret_cast<const float4 *>(xa)[0], x47 = reinterpret_cast<const float4 *>(xa)[1];
                        const int nyp = __builtin_bit_cast(int, y47.w), nxp = __builtin_bit_cast(int, x47.w);
                        const float wys[kTlTaps] = {y03.x, y03.y, y03.z, y03.w, y47.x, y47.y};
                        const float wxs[kTlTaps] = {x03.x, x03.y, x03.z, x03.w, x47.x, x47.y};
                        const int base = (__builtin_bit_cast(int, y47.z) - ry0) * rw + (__builtin_bit_cast(int, x47.z) - rx0);
                        const int last = npx - 1;
#pragma unroll
                        for (int ky = 0; ky < kTlTaps; ky++) {
                            if (ky < ty) {                              // (wave-uniform bound: no divergence, static register indices)
#pragma unroll
                                for (int kx = 0; kx < kTlTaps; kx++) {
                                    if (kx < tx) {
                                        const bool on = ky < nyp && kx < nxp;
                                        const float4 t = win[max(0, min(base + ky * rw + kx, last)) * kTlQN + q];
                                        const float w = wys[ky] * wxs[kx];
                                        acc.x = on ? fmaf(w, t.x, acc.x) : acc.x;
                                        acc.y = on ? fmaf(w, t.y, acc.y) : acc.y;
                                        acc.z = on ? fmaf(w, t.z, acc.z) : acc.z;
                                        acc.w = on ? fmaf(w, t.w, acc.w) : acc.w;
                                    }
                                }
                            }
                        }
                    }
                    float *t = tile + (4 * q) * ts + ph * PW + pw;
                    t[0] = acc.x * inv_count;
                    t[ts] = acc.y * inv_count;
                    t[2 * ts] = acc.z * inv_count;
                    t[3 * ts] = acc.w * inv_count;
                }
            }
        }
    } else {
        // The exact arithmetic for a ROI the plan does not cover (and zeros for an invalid batch index): torchvision's order,
        // un-fused, samples computed on the fly -- rare by construction, so no tables.
        const float *roi = rois + r * 5;
        const float off = aligned ? 0.5f : 0.0f;
        const float start_w = roi[1] * scale - off, start_h = roi[2] * scale - off;
        const float end_w = roi[3] * scale - off, end_h = roi[4] * scale - off;
        float rw = end_w - start_w, rh = end_h - start_h;
        if (!aligned) {
            rw = fmaxf(rw, 1.f);
            rh = fmaxf(rh, 1.f);
        }
        const float bin_h = rh / (float)PH, bin_w = rw / (float)PW;
        int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(bin_h);
        int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(bin_w);
        const int prod = gh * gw;
        const float count = (float)(prod > 1 ? prod : 1);
        gh = (gh > 0 && batch >= 0) ? gh : 0;
        gw = (gw > 0 && batch >= 0) ? gw : 0;
        const unsigned ch_off = (unsigned)(c_ok ? cq : 0) * (unsigned)sizeof(float);
        for (int g0 = 0; g0 < bins; g0 += kTlThreads / kTlQN) {
            const int bin = g0 + wave * (64 / kTlQN) + sub;
            if (bin >= bins) continue;
            const int ph = (int)(((float)bin + 0.5f) * inv_pw), pw = bin - ph * PW;
            float4 acc = {0.f, 0.f, 0.f, 0.f};
            if (c_ok) {
                for (int iy = 0; iy < gh; iy++) {
                    const AxisSampleN ys = axis_sample_n(start_h, bin_h, ph, iy, gh, H);
                    for (int ix = 0; ix < gw; ix++) {
                        const AxisSampleN xs = axis_sample_n(start_w, bin_w, pw, ix, gw, W);
                        const unsigned ylo = (unsigned)ys.lo * ystride + ch_off, yhi = (unsigned)ys.hi * ystride + ch_off;
                        const unsigned xlo = (unsigned)xs.lo * xstride, xhi = (unsigned)xs.hi * xstride;
                        const float w1 = ys.wh * xs.wh, w2 = ys.wh * xs.wl, w3 = ys.wl * xs.wh, w4 = ys.wl * xs.wl;
                        const float4 v1 = tap(ylo + xlo), v2 = tap(ylo + xhi), v3 = tap(yhi + xlo), v4 = tap(yhi + xhi);
                        acc.x = acc.x + (((w1 * v1.x + w2 * v2.x) + w3 * v3.x) + w4 * v4.x);
                        acc.y = acc.y + (((w1 * v1.y + w2 * v2.y) + w3 * v3.y) + w4 * v4.y);
                        acc.z = acc.z + (((w1 * v1.z + w2 * v2.z) + w3 * v3.z) + w4 * v4.z);
                        acc.w = acc.w + (((w1 * v1.w + w2 * v2.w) + w3 * v3.w) + w4 * v4.w);
                    }
                }
            }
            float *t = tile + (4 * q) * ts + bin;
            t[0] = acc.x / count;
            t[ts] = acc.y / count;
            t[2 * ts] = acc.z / count;
            t[3 * ts] = acc.w / count;
        }
    }
    __syncthreads();
    const int cn = min(kTlCh, C - c0);
    float *dst = out + (r * C + c0) * (int64_t)bins;
    if ((bins & 3) == 0) {
        // 16 bytes per lane: four consecutive bins of one channel; (channel, bin quad) advance incrementally
        const int qpc = bins >> 2;
        int c = 0, b4 = tid;
        while (b4 >= qpc) {
            b4 -= qpc;
            c++;
        }
        const int step_c = kTlThreads / qpc, step_b = kTlThreads - step_c * qpc;
        while (c < cn) {
            const float *t = tile + c * ts + 4 * b4;
            const float4 v = {t[0], t[1], t[2], t[3]};
            *reinterpret_cast<float4 *>(dst + (c * bins + 4 * b4)) = v;
            c += step_c;
            b4 += step_b;
            if (b4 >= qpc) {
                b4 -= qpc;
                c++;
            }
        }
        return;
    }
    const float inv_bins = 1.0f / (float)bins;
    for (int idx = tid; idx < cn * bins; idx += kTlThreads) {
        const int c = (int)(((float)idx + 0.5f) * inv_bins);
        dst[idx] = tile[c * ts + (idx - c * bins)];
    }
}

int64_t roi_align_tiles_plan_bytes(int64_t R) { return R * (int64_t)sizeof(RoiPlan); }

int launch_roi_align_tiles(const float *feat_nhwc, int N, int H, int W, int C, const float *rois, int64_t R, int PH, int PW,
                           float scale, int sampling_ratio, int aligned, void *plan_ws, float *out, hipStream_t s)
{
    const int bins = PH * PW, ts = bins | 1;
    const size_t lds = ((size_t)kTlCh * ts + 4) * sizeof(float) + (size_t)kTlWinPix * kTlCh * sizeof(float) + sizeof(RoiPlan);
    if (lds > 150 * 1024) return set_error(LOCOV_ERR_UNSUPPORTED, "locov_roi_align_from_nhwc_fwd: pooled size %dx%d too large for the LDS tile", PH, PW);
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(roi_align_tiles_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return set_error(LOCOV_ERR_LAUNCH, "locov_roi_align_from_nhwc_fwd: cannot raise the dynamic LDS limit to %zu bytes", lds);
    RoiPlan *plans = static_cast<RoiPlan *>(plan_ws);
    hipLaunchKernelGGL(roi_plan_kernel, dim3((unsigned)ceil_div(R, 64)), dim3(64), 0, s, rois, R, N, H, W, PH, PW, scale, sampling_ratio,
                       aligned, plans);
    int rc = check_launch("locov_roi_align_from_nhwc_fwd (plan)");
    if (rc) return rc;
    hipLaunchKernelGGL(roi_align_tiles_kernel, dim3((unsigned)R, (unsigned)ceil_div(C, kTlCh)), dim3(kTlThreads), lds, s, feat_nhwc, N, H,
                       W, C, rois, plans, PH, PW, scale, sampling_ratio, aligned, out);
    return check_launch("locov_roi_align_from_nhwc_fwd (tiles)");
}

}  // namespace locov
