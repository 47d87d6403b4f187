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
#include "roi_align_common.h"
#include "winograd_transform.h"

#include <cstdlib>
#include <type_traits>

namespace locov {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 load4(const __bf16 *p)
{
    const bf16x4 v = *reinterpret_cast<const bf16x4 *>(p);
    return float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void store4(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }
// The pooled rows are consumed by a LATER kernel (GBs of other traffic in between), while the map slice they were gathered from
// is re-read by every ROI of the image: `sc1` stores leave no copy of the written line in the XCD's L2 (MI355X_MICROARCH.md,
// stores of each flavour), so the 3.2 GB of output no longer push the 2-4 MB map slice out of it.
// Measured on the 2 048-channel launch of block 0's shortcut (8 x 1000 proposals; tools/ab_pool.sh, docs/experiments.md R4):
// plain stores 1.17 ms with 3.9 GB of fabric reads for a 0.28 GB map; `sc1` 1.18 ms / 1.6 GB; `nt` 1.06 ms / 1.8 GB -- nt it is.
#ifndef LOCOV_POOL_STORE_AUX
#define LOCOV_POOL_STORE_AUX 2                             // 0 = plain, 2 = nt, 16 = sc1, 17 = sc0 sc1 (developer A/B)
#endif
#ifndef LOCOV_T2_STORE_AUX
#define LOCOV_T2_STORE_AUX 2                               // the pooler-contract kernel's NCHW stores: nt 2.88 ms, plain 2.94, sc1 2.98 (tools/ab_t2.py)
#endif
#ifndef LOCOV_POOLWINO_NT
#define LOCOV_POOLWINO_NT 0                                // the WINO pooler's transform-domain stores (developer A/B)
#endif
template <int AUX>
__device__ __forceinline__ void store4_policy(float *p, const float4 &v)
{
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const f32x4_t d = {v.x, v.y, v.z, v.w};
    if (AUX == 16) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(d) : "memory");
    else if (AUX == 17) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(d) : "memory");
    else if (AUX == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(d) : "memory");
    else *reinterpret_cast<float4 *>(p) = v;
}
#ifndef LOCOV_POOL_NO_STORE
#define LOCOV_POOL_NO_STORE 0                              // developer timing (tools/ab_pool_nostore.sh): 1 = the gather alone (wrong results)
#endif
__device__ __forceinline__ void store4_out(float *p, const float4 &v)
{
    if (LOCOV_POOL_NO_STORE && v.x != 1234.56789f) return;       // (a value nothing takes: the loads stay, the store goes)
    store4_policy<LOCOV_POOL_STORE_AUX>(p, v);
}
__device__ __forceinline__ void store4(__bf16 *p, const float4 &v)
{
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4 *>(p) = o;
}
__device__ __forceinline__ void store4_out(__bf16 *p, const float4 &v) { store4(p, v); }

// one tap = 4 consecutive channels through a raw buffer descriptor (byte offset in a VGPR, base in SGPRs)
__device__ __forceinline__ float4 tap4(__amdgpu_buffer_rsrc_t r, unsigned off, float *)
{
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float4 tap4(__amdgpu_buffer_rsrc_t r, unsigned off, __bf16 *)
{
    const bf16x4 v = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    return float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

constexpr int kNhwcThreads = 256;
constexpr int kMaxAxisN = 192;     // per-axis LDS table entries (7 x up to 27 samples; larger grids are computed on the fly): keeps the kernel at 6+ workgroups per CU

// grid = R * nslices (1-D): one workgroup = one ROI x one CHANNEL SLICE, all OH x OW bins.
//
// Why slices: workgroups are dealt round-robin over the 8 XCDs, each with its own 4 MB L2.  With one workgroup per
// (ROI, bin row) and all channels, every XCD touched every channel of every image: on the map path (2 560 pooled channels,
// 43 MB per 1333x800 image) each bin's pixels came from beyond L2 -- 11.2 GB of fabric reads per launch pair for 4 GB of
// output (rocprofv3 FETCH_SIZE, profiles/r01l).  Now blockIdx % nslices selects the slice, i.e. (nslices = 8) the XCD: an
// XCD only ever reads ITS slice of the channels -- 4 200 pixels x C/8 channels of the image being pooled, 1.3 MB (512
// channels) to 5.4 MB (2 560) -- and consecutive ROIs (same image) run back to back on it, so the footprints of an image's
// proposals, which overlap heavily, are served by that XCD's L2.  The per-ROI sampling tables are built once per workgroup
// for all OH bin rows (they used to be rebuilt per bin row).
// BWD = true is the adjoint with the same sampling geometry: `out` then holds the GRADIENT of the pooled rows (read) and
// `feat` the gradient of the channels-last map (accumulated with fp32 hardware atomics; the caller zeroes it) -- what
// autograd needs when the LSM head trains through the even-grid pooler (roi_emb_heads.py:343 under autograd).
//
// WINO = true (fp32, 7 x 7 strided bins, slices of at most 64 channels): the pooled + FrozenBN + ReLU values are the input of a 3x3
// convolution evaluated in the Winograd domain (block 0's conv2).  They are kept in LDS ([49][64]) instead of being stored, and
// the workgroup writes their input transform V [121][R][C] (`out`, split layout x v_scale) itself -- wino_in_fy of
// winograd_transform.h, the bits wino_input_kernel<false, true> would have produced from the stored rows.
// WINO = the slice width in channels (0: not the WINO form).  128 where C allows it: the transform phase's lane = channel pair then
// fills its waves (64 pairs), a wave's store is a 512-byte run, and a 512-channel map makes 4 slices (two XCDs share one: 2.15 MB per
// image, still inside L2) -- block 0's pooler + conv2 2.73 -> 2.60 ms at 8 000 proposals (tools/attic/dbg_fuse_pool.py); 64 otherwise.
template <typename TIn, typename TOut, bool BWD = false, int WINO = 0>
__global__ __launch_bounds__(kNhwcThreads, WINO == 64 ? 6 : WINO ? 4 : 1) void roi_align_nhwc_kernel(
    const TIn *__restrict__ feat, int N, int H, int W, int C, const float *__restrict__ rois, int PH, int PW,
    float scale, int sampling_ratio, int aligned, int bin_stride, int OH, int OW, int pos_major,
    TOut *__restrict__ out, int64_t out_ld, int64_t feat_ld, const float *__restrict__ ch_scale,
    const float *__restrict__ ch_shift, int relu, int nslices, int64_t R, float v_scale = 1.f, unsigned *overflow = nullptr,
    int bwd_win_floats = 0)
{
    // BWD: dynamic LDS of bwd_win_floats floats -- the gradient window of a SMALL proposal (see the BWD branch below)
    extern __shared__ float bwd_win[];
    constexpr int kWinoPitch = WINO + 4;
    __shared__ float wino_tile_s[WINO ? 49 * kWinoPitch : 1];
    float *const wino_tile = wino_tile_s;
    // feat_ld = elements between consecutive pixels of the map (>= C: the C channels may be a column block
    // of a wider per-pixel vector).  ch_scale / ch_shift / relu: optional per-channel affine + ReLU applied to
    // the pooled value -- ROIAlign is linear, so a 1x1 convolution can run on the MAP (once per pixel instead
    // of once per ROI bin) and its FrozenBN + ReLU are applied here, after the pooling.
    __shared__ AxisSampleN ytab[kMaxAxisN];
    __shared__ AxisSampleN xtab[kMaxAxisN];

    // up to 8 slices: blockIdx % nslices = the slice = (round-robin dispatch) the XCD.  More than 8 (developer A/B,
    // LOCOV_ROIALIGN_SLICES): passes of 8 slices, every ROI of pass p before any of pass p + 1, so that an XCD still works on ONE
    // slice at a time
    // slice at a time.  The LAST pass may be ragged (nslices = 9, 12, ...: C = 576, 1536 on 64- / 128-channel slices): it holds
    // the remaining nslices - 8 * pass slices, every ROI of each
    const unsigned per = nslices > 8 ? 8u : (unsigned)nslices, per_pass = per * (unsigned)R;
    const unsigned pass = blockIdx.x / per_pass, rem = blockIdx.x - pass * per_pass;
    const unsigned left = (unsigned)nslices - pass * per, per_here = left < per ? left : per;
    const int slice = (int)(rem % per_here + pass * per);
    const int64_t r = rem / per_here;
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
    // tables of every (strided) bin row's y samples and of every (strided) column's x samples
    const int ny = OH * gh, nx = OW * gw;
    const bool use_lds = ny <= kMaxAxisN && nx <= kMaxAxisN;
    // (the tables hold BYTE offsets into the image -- row offset for y, pixel offset for x -- so that a tap
    // address is two 32-bit adds on top of a wave-uniform buffer descriptor)
    const unsigned xstride = (unsigned)feat_ld * (unsigned)sizeof(TIn), ystride = (unsigned)W * xstride;
    auto as_offsets = [](AxisSampleN a, unsigned stride) {
        a.lo = (int)((unsigned)a.lo * stride);
        a.hi = (int)((unsigned)a.hi * stride);
        return a;
    };
    if (use_lds) {
        for (int t = threadIdx.x; t < ny; t += kNhwcThreads)
            ytab[t] = as_offsets(axis_sample_n(start_h, bin_h, (t / gh) * bin_stride, t % gh, gh, H), ystride);
        for (int t = threadIdx.x; t < nx; t += kNhwcThreads)
            xtab[t] = as_offsets(axis_sample_n(start_w, bin_w, (t / gw) * bin_stride, t % gw, gw, W), xstride);
    }
    __syncthreads();

    // Separable form.  The bin value is sum_samples sum_taps wy*wx*F = sum_{pixel rows} sum_{pixel cols} Wy[y] Wx[x] F[y][x]
    // with Wy / Wx the per-PIXEL sums of the samples' bilinear weights.  Samples are at most one pixel apart
    // (grid = ceil(bin size)), so a bin touches at most (gh+1) x (gw+1) distinct pixels instead of 4*gh*gw taps:
    // 9 instead of 16 loads at a 2x2 grid, 25 instead of 64 at 4x4.  Same sum, re-associated (fp32 rounding only).
    constexpr int kSepGrid = 16, kSepCols = 32;
    __shared__ float ypw[kSepCols * (kSepGrid + 1)];
    __shared__ float xpw[kSepCols * (kSepGrid + 1)];
    __shared__ int ypix[kSepCols][2];             // per bin row: {byte offset of the first pixel row, number of rows}
    __shared__ int xpix[kSepCols][2];
    __shared__ int sep_bad;
    const bool sep_try = use_lds && gh >= 1 && gw >= 1 && gh <= kSepGrid && gw <= kSepGrid && OW <= kSepCols && OH <= kSepCols;
    if (threadIdx.x == 0) sep_bad = 0;
    __syncthreads();
    if (sep_try && (int)threadIdx.x < OH + OW) {
        // thread oh: the y axis of bin row oh; thread OH + ow: the x axis of output column ow
        const bool is_y = (int)threadIdx.x < OH;
        const int idx_t = is_y ? (int)threadIdx.x : (int)threadIdx.x - OH, n = is_y ? gh : gw;
        const AxisSampleN *tab = is_y ? ytab + idx_t * gh : xtab + idx_t * gw;
        const unsigned stride = is_y ? ystride : xstride;
        float *pw = (is_y ? ypw : xpw) + idx_t * (kSepGrid + 1);
        int base = 0x7fffffff;
        for (int t = 0; t < n; t++)
            if (tab[t].wl != 0.f || tab[t].wh != 0.f) base = min(base, tab[t].lo);
        int num = 0;
        if (base != 0x7fffffff) {
            for (int k = 0; k <= n; k++) pw[k] = 0.f;
            for (int t = 0; t < n; t++) {
                const AxisSampleN sm = tab[t];
                if (sm.wl == 0.f && sm.wh == 0.f) continue;
                const int klo = (int)((unsigned)(sm.lo - base) / stride), khi = (int)((unsigned)(sm.hi - base) / stride);
                if (khi > n) {
                    sep_bad = 1;
                    break;
                }
                pw[klo] += sm.wh;
                pw[khi] += sm.wl;
                num = max(num, khi + 1);
            }
        }
        int (*pix)[2] = is_y ? ypix : xpix;
        pix[idx_t][0] = base == 0x7fffffff ? 0 : base;
        pix[idx_t][1] = num;
    }
    __syncthreads();
    const bool separable = sep_try && !sep_bad;

    // this workgroup's channel slice [c_lo, c_hi): multiples of 4 channels
    const int c4_all = C >> 2, c4s = (c4_all + nslices - 1) / nslices;
    const int q_lo = slice * c4s, q_hi = min(q_lo + c4s, c4_all);
    // (BWD works on single channels, not quads: a wave's atomic instruction then covers 256 contiguous bytes = four full
    // 64-byte memory-side atomic requests instead of sixteen quarter-used ones)
    const int c4n = BWD ? 4 * (q_hi - q_lo) : q_hi - q_lo;
    if (c4n <= 0) return;
    const bool valid_b = b >= 0 && b < N;
    const TIn *img = feat + (int64_t)(valid_b ? b : 0) * H * W * feat_ld;
    const __amdgpu_buffer_rsrc_t img_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<TIn *>(img), 0, (unsigned)H * ystride, 0x00020000);
    // ROI-major: out[r][oh][ow][c]; position-major: out[oh][ow][r][c] (R rows per position)
    // (out_ld = elements between consecutive pixel rows, >= C: the rows may be a column block of a wider matrix)
    const int64_t bin_stride_out = pos_major ? R * out_ld : out_ld;
    TOut *obase = pos_major ? out + r * out_ld : out + (r * OH * (int64_t)OW) * out_ld;
    const int nbins = OH * OW;
    if constexpr (BWD) {
        // SMALL proposals (the 49 bins of a proposal below ~70 px land on at most 40 distinct map pixels, 4 pixels each): the bins'
        // contributions are first added up per pixel in an LDS window over the proposal's pixel rectangle (ds_add_f32), then every
        // touched (pixel, channel) costs ONE memory-side atomic instead of five to fifty on the same line.  The scatter of the LARGE
        // proposals, which bounds the launch, gets the atomic units the small ones no longer occupy: LSM step's launch 0.77 -> 0.60 ms,
        // STT's 1.44 -> 1.08 ms (the launcher's comment has the window-size sweep).
        __shared__ int rect[4];                            // {y0 (byte offset), x0 (byte offset), rows, columns}
        if (threadIdx.x == 0) {
            int ya = 0x7fffffff, yb = -1, xa = 0x7fffffff, xb = -1;
            if (separable && valid_b) {
                for (int i = 0; i < OH; i++)
                    if (ypix[i][1] > 0) {
                        ya = min(ya, ypix[i][0]);
                        yb = max(yb, ypix[i][0] + (ypix[i][1] - 1) * (int)ystride);
                    }
                for (int i = 0; i < OW; i++)
                    if (xpix[i][1] > 0) {
                        xa = min(xa, xpix[i][0]);
                        xb = max(xb, xpix[i][0] + (xpix[i][1] - 1) * (int)xstride);
                    }
            }
            const bool any = yb >= 0 && xb >= 0;
            rect[0] = ya;
            rect[1] = xa;
            rect[2] = any ? (yb - ya) / (int)ystride + 1 : 0;
            rect[3] = any ? (xb - xa) / (int)xstride + 1 : 0;
        }
        __syncthreads();
        const int wh = rect[2], ww = rect[3];
        if (wh > 0 && (int64_t)wh * ww * c4n <= bwd_win_floats) {
            const int y0 = rect[0], x0 = rect[1];
            const int nwin = wh * ww * c4n;
            for (int i = threadIdx.x; i < nwin; i += kNhwcThreads) bwd_win[i] = 0.f;
            __syncthreads();
            // thread = (bin, channel), bins advanced incrementally (c4n channels per bin)
            int bin = 0, oh = 0, ow = 0, cq = threadIdx.x;
            while (true) {
                while (cq >= c4n) {
                    cq -= c4n;
                    bin++;
                    if (++ow == OW) {
                        ow = 0;
                        oh++;
                    }
                }
                if (bin >= nbins) break;
                const TOut *optr = obase + (int64_t)bin * bin_stride_out + ((q_lo << 2) + cq);
                const float g = (float)optr[0] * inv_count;
                const int nyp = ypix[oh][1], nxp = xpix[ow][1];
                const float *yw = ypw + oh * (kSepGrid + 1), *xw = xpw + ow * (kSepGrid + 1);
                const int py0 = (ypix[oh][0] - y0) / (int)ystride, px0 = (xpix[ow][0] - x0) / (int)xstride;
                for (int ky = 0; ky < nyp; ky++)
                    for (int kx = 0; kx < nxp; kx++) {
                        const float w = yw[ky] * xw[kx];
                        if (w != 0.f)      // (the native LDS float add, ds_add_f32: the generic atomicAdd compiles to a compare-and-swap loop)
                            __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)&bwd_win[((py0 + ky) * ww + px0 + kx) * c4n + cq],
                                                      w * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
                    }
                cq += kNhwcThreads;
            }
            __syncthreads();
            char *gimg = reinterpret_cast<char *>(const_cast<TIn *>(img));
            // flush: consecutive threads = consecutive channels of one pixel (256-byte atomic instructions per wave)
            int pix = 0, c = threadIdx.x;
            while (true) {
                while (c >= c4n) {
                    c -= c4n;
                    pix++;
                }
                if (pix >= wh * ww) break;
                const float v = bwd_win[pix * c4n + c];
                if (v != 0.f) {
                    const int py = pix / ww, px = pix - py * ww;
                    unsafeAtomicAdd(reinterpret_cast<float *>(gimg + (unsigned)y0 + (unsigned)py * ystride + (unsigned)x0 + (unsigned)px * xstride +
                                                              (unsigned)((q_lo << 2) + c) * (unsigned)sizeof(TIn)),
                                    v);
                }
                c += kNhwcThreads;
            }
            return;
        }
    }
    // (bin, channel quad) of this thread: advanced incrementally, no integer division in the loop
    int bin = 0, oh = 0, ow = 0, cq = threadIdx.x;
    auto normalise = [&]() {
        while (cq >= c4n) {
            cq -= c4n;
            bin++;
            if (++ow == OW) {
                ow = 0;
                oh++;
            }
        }
    };
    normalise();
    while (bin < nbins) {
        const int c = BWD ? (q_lo << 2) + cq : (q_lo + cq) << 2;
        const unsigned ch_off = (unsigned)c * (unsigned)sizeof(TIn);
        TOut *optr = obase + (int64_t)bin * bin_stride_out + c;
        const float *yw = ypw + oh * (kSepGrid + 1);
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BWD) {
            if (valid_b) {
                const float g = (float)optr[0] * inv_count;
                char *gimg = reinterpret_cast<char *>(const_cast<TIn *>(img));
                auto scatter = [&](unsigned off, float w) {
                    if (w == 0.f) return;
                    unsafeAtomicAdd(reinterpret_cast<float *>(gimg + off), w * g);
                };
                if (separable) {
                    const int nyp = ypix[oh][1], nxp = xpix[ow][1];
                    const unsigned y0 = (unsigned)ypix[oh][0] + (unsigned)xpix[ow][0] + ch_off;
                    const float *xw = xpw + ow * (kSepGrid + 1);
                    for (int ky = 0; ky < nyp; ky++)
                        for (int kx = 0; kx < nxp; kx++) scatter(y0 + (unsigned)ky * ystride + (unsigned)kx * xstride, yw[ky] * xw[kx]);
                } else {
                    for (int iy = 0; iy < gh; iy++) {
                        const AxisSampleN ys = use_lds ? ytab[oh * gh + iy]
                                                       : as_offsets(axis_sample_n(start_h, bin_h, oh * bin_stride, iy, gh, H), ystride);
                        for (int ix = 0; ix < gw; ix++) {
                            const AxisSampleN xs = use_lds ? xtab[ow * gw + ix]
                                                           : as_offsets(axis_sample_n(start_w, bin_w, ow * bin_stride, ix, gw, W), xstride);
                            scatter((unsigned)ys.lo + (unsigned)xs.lo + ch_off, ys.wh * xs.wh);
                            scatter((unsigned)ys.lo + (unsigned)xs.hi + ch_off, ys.wh * xs.wl);
                            scatter((unsigned)ys.hi + (unsigned)xs.lo + ch_off, ys.wl * xs.wh);
                            scatter((unsigned)ys.hi + (unsigned)xs.hi + ch_off, ys.wl * xs.wl);
                        }
                    }
                }
            }
            cq += kNhwcThreads;
            normalise();
            continue;
        }
        if (valid_b && separable) {
            const int nyp = ypix[oh][1], nxp = xpix[ow][1];
            const unsigned x0 = (unsigned)xpix[ow][0] + ch_off;
            const float *xw = xpw + ow * (kSepGrid + 1);
            // the gather is latency-bound: up to 8 pixels are in flight per lane before any is consumed
            // (pixel counters are wave-uniform -> scalar registers)
            const unsigned y0 = (unsigned)ypix[oh][0] + x0;
            const int npix = nyp * nxp;
            int ky = 0, kx = 0;
            auto group = [&](auto nu_tag) __attribute__((always_inline)) {
                constexpr int NU = decltype(nu_tag)::value;
                float4 v[NU];
                float wgt[NU];
#pragma unroll
                for (int u = 0; u < NU; u++) {
                    wgt[u] = yw[ky] * xw[kx];
                    v[u] = tap4(img_rsrc, y0 + (unsigned)ky * ystride + (unsigned)kx * xstride, (TIn *)nullptr);
                    if (++kx == nxp) {
                        kx = 0;
                        ky++;
                    }
                }
#pragma unroll
                for (int u = 0; u < NU; u++) {
                    acc.x = fmaf(wgt[u], v[u].x, acc.x);
                    acc.y = fmaf(wgt[u], v[u].y, acc.y);
                    acc.z = fmaf(wgt[u], v[u].z, acc.z);
                    acc.w = fmaf(wgt[u], v[u].w, acc.w);
                }
            };
            int p = 0;
            for (; p + 8 <= npix; p += 8) group(std::integral_constant<int, 8>{});
            const int rem = npix - p;                            // 0..7: binary decomposition
            if (rem & 4) group(std::integral_constant<int, 4>{});
            if (rem & 2) group(std::integral_constant<int, 2>{});
            if (rem & 1) group(std::integral_constant<int, 1>{});
        } else if (valid_b) {
            for (int iy = 0; iy < gh; iy++) {
                const AxisSampleN ys = use_lds ? ytab[oh * gh + iy]
                                               : as_offsets(axis_sample_n(start_h, bin_h, oh * bin_stride, iy, gh, H), ystride);
                const unsigned ylo = (unsigned)ys.lo + ch_off, yhi = (unsigned)ys.hi + ch_off;
                for (int ix = 0; ix < gw; ix++) {
                    const AxisSampleN xs = use_lds ? xtab[ow * gw + ix]
                                                   : as_offsets(axis_sample_n(start_w, bin_w, ow * bin_stride, ix, gw, W), xstride);
                    const float w1 = ys.wh * xs.wh, w2 = ys.wh * xs.wl, w3 = ys.wl * xs.wh, w4 = ys.wl * xs.wl;
                    const float4 v1 = tap4(img_rsrc, ylo + (unsigned)xs.lo, (TIn *)nullptr);
                    const float4 v2 = tap4(img_rsrc, ylo + (unsigned)xs.hi, (TIn *)nullptr);
                    const float4 v3 = tap4(img_rsrc, yhi + (unsigned)xs.lo, (TIn *)nullptr);
                    const float4 v4 = tap4(img_rsrc, yhi + (unsigned)xs.hi, (TIn *)nullptr);
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
        if (ch_scale) {
            const float4 sc = *reinterpret_cast<const float4 *>(ch_scale + c);
            acc.x *= sc.x; acc.y *= sc.y; acc.z *= sc.z; acc.w *= sc.w;
        }
        if (ch_shift) {
            const float4 sh = *reinterpret_cast<const float4 *>(ch_shift + c);
            acc.x += sh.x; acc.y += sh.y; acc.z += sh.z; acc.w += sh.w;
        }
        if (relu) {
            acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
        }
        if constexpr (WINO)
            *reinterpret_cast<float4 *>(wino_tile + bin * kWinoPitch + 4 * cq) = acc;
        else
            store4_out(optr, acc);
        cq += kNhwcThreads;
        normalise();
    }
    if constexpr (WINO && !BWD && std::is_same<TOut, float>::value) {
        __syncthreads();
        // wave w takes the rows fy = w, w + 4, w + 8 of the transform; lane = channel pair of the slice (a slice has an even
        // number of pairs: lanes trade words in pairs)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int npairs = 2 * c4n;
        float amax = 0.f;
        if (lane < npairs) {
            const int c = (q_lo << 2) + 2 * lane;
            const float *patch = wino_tile + 2 * lane;
            const int64_t fstride = R * C;
            float *vrow = reinterpret_cast<float *>(out) + r * C;
            auto load = [&](int y, int xx) __attribute__((always_inline)) {
                return *reinterpret_cast<const f32x2 *>(patch + (y * 7 + xx) * kWinoPitch);
            };
            auto unit = [&](auto fy_tag) __attribute__((always_inline)) {
                constexpr int FY = decltype(fy_tag)::value;
                wino_in_fy<false, FY>(load, [&](int fx, f32x2 a) __attribute__((always_inline)) {
                    amax = fmaxf(fmaxf(amax, fabsf(a[0])), fabsf(a[1]));
                    store_split_pair_policy<LOCOV_POOLWINO_NT != 0>(vrow + (int64_t)(FY * wino::NF + fx) * fstride, c, a, v_scale);
                });
                __builtin_amdgcn_sched_barrier(0);             // one row of the transform at a time: ~80 live registers, not 11 rows' worth
            };
            using std::integral_constant;
            if (wave == 0) {
                unit(integral_constant<int, 0>{}); unit(integral_constant<int, 4>{}); unit(integral_constant<int, 8>{});
            } else if (wave == 1) {
                unit(integral_constant<int, 1>{}); unit(integral_constant<int, 5>{}); unit(integral_constant<int, 9>{});
            } else if (wave == 2) {
                unit(integral_constant<int, 2>{}); unit(integral_constant<int, 6>{}); unit(integral_constant<int, 10>{});
            } else {
                unit(integral_constant<int, 3>{}); unit(integral_constant<int, 7>{});
            }
        }
        if (overflow != nullptr && amax * v_scale >= 65504.f) atomicOr(overflow, 1u);
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

// ---------------------------------------------------------------------------------------------
// The pooler contract ([R,C,ph,pw] out of an NCHW map) at HBM speed: gather from a channels-last
// COPY of the map, transpose in LDS, write NCHW.
//
// Gathering from NCHW costs one scattered dword per lane and tap (5-10 cache lines per wave
// instruction: the texture-address path, not HBM, bounds the kernel at ~0.6 TB/s).  From a
// channels-last copy a tap is one contiguous C-vector: 16 lanes x 16 B cover 64 channels, a wave
// covers several bins per instruction.  One workgroup = one ROI x kT2Ch channels: results go to an
// LDS tile [ch][ph*pw] (odd row stride) and leave as one contiguous, fully coalesced run of
// kT2Ch*ph*pw floats -- exactly the layout of out[r, c0:c0+kT2Ch, :, :].
// Arithmetic and summation order per output element are those of roi_align_nchw_kernel (and of
// the oracle): the result is bit-identical.
// ---------------------------------------------------------------------------------------------
constexpr int kT2Threads = 256;
constexpr int kT2Ch = 32;          // channels per workgroup (one 128-byte line per tap; ~25 KiB LDS tile)
constexpr int kT2Axis = 256;       // per-axis LDS table entries (larger sampling grids: computed on the fly)

// ---- small proposals: the whole footprint in LDS (roi_align_win_kernel) --------------------------------------------------
// A proposal of up to ~11 x 11 map pixels (side <= ~180 image pixels: 6 of 10 bench proposals) touches at most 14 x 14 pixels, and
// the [32 ch][bins] transpose tile is 25 KB = 197 pixels x 128 B.  For those ROIs the workgroup fetches the pixel rectangle ONCE
// (<= 7 loads per thread, all in flight: one memory latency instead of one per pass of 32 bins), takes every tap from LDS -- the
// same values in the same per-sample order: bit-identical -- keeps its 7 results per thread in registers and only then re-uses
// the same LDS bytes as the transpose tile.  LDS per workgroup and waves per CU are those of the direct form (what sank round 3's
// LDS-staged kernel and round 4's pipelined one was occupancy).  roi_window_rect decides per workgroup, from the proposal's coordinates
// alone, which path it takes.
constexpr int kWinPitch = 4 * kT2Ch;                      // bytes per pixel of the window (32 channels)
constexpr int kWinMaxPasses = 7;                          // results per thread kept in registers: bins <= 7 * 32
#ifndef LOCOV_ROIALIGN_WINDOW
#define LOCOV_ROIALIGN_WINDOW 1                           // developer A/B: 0 = every proposal takes the direct form
#endif

// conservative pixel rectangle of every tap of the ROI: one pixel of margin on each side absorbs the difference between this
// estimate's rounding and axis_sample_n's.  Returns false when it does not fit `cap` pixels (or the ROI cannot use the window).
__device__ __forceinline__ bool roi_window_rect(float start_h, float start_w, float bin_h, float bin_w, int gh, int gw, int PH, int PW, int H,
                                                int W, int cap, bool valid_b, int &y0, int &x0, int &wh, int &ww)
{
    if (!LOCOV_ROIALIGN_WINDOW || !valid_b || gh <= 0 || gw <= 0 || PH * PW > kWinMaxPasses * (kT2Threads / (kT2Ch / 4)) || PH * gh > kT2Axis ||
        PW * gw > kT2Axis)
        return false;
    // every sample lies between the ROI's two edges (an inverted ROI under a forced sampling ratio runs from the far edge back)
    const float ya_ = start_h, yb_ = start_h + (float)PH * bin_h, xa_ = start_w, xb_ = start_w + (float)PW * bin_w;
    const float yf = fminf(ya_, yb_), yl = fmaxf(ya_, yb_), xf = fminf(xa_, xb_), xl = fmaxf(xa_, xb_);
    if (!(yl - yf < 64.f) || !(xl - xf < 64.f) || !(yf > -1.0e6f) || !(xf > -1.0e6f) || !(yl < 1.0e6f) || !(xl < 1.0e6f)) return false;   // (also rejects NaN)
    const int ya = max((int)floorf(fmaxf(yf, 0.f)) - 1, 0), yb = min((int)floorf(fmaxf(yl, 0.f)) + 2, H - 1);
    const int xa = max((int)floorf(fmaxf(xf, 0.f)) - 1, 0), xb = min((int)floorf(fmaxf(xl, 0.f)) + 2, W - 1);
    y0 = min(ya, H - 1);
    x0 = min(xa, W - 1);
    wh = max(yb - y0 + 1, 1);
    ww = max(xb - x0 + 1, 1);
    // (the window path stages at most kWinMaxPasses pixels per thread group: a tile of more than 224 floats per channel row could hold more)
    return wh * ww <= min(cap, kWinMaxPasses * (kT2Threads / (kT2Ch / 4)));
}

// (a device function of roi_align_nhwc2nchw_kernel, not a launch of its own: as two launches the direct form's texture-bound large
//  proposals and the window form's store-bound small ones ran one after the other instead of beside each other -- 3.05 ms against
//  2.84; the two paths share the kernel's register allocation, the larger of the two)
__device__ __forceinline__ void roi_align_window_path(
    const float *__restrict__ feat, int N, int H, int W, int C, const float *__restrict__ rois, int PH, int PW,
    float scale, int sampling_ratio, int aligned, float *__restrict__ out, float *smem, int y0, int x0, int wh, int ww)
{
    const int bins = PH * PW;
    const int ts = bins | 1;                                  // odd row stride of the transpose tile
    float *tile = smem;                                       // [kT2Ch][ts] -- first the pixel window, then the tile
    char *win = reinterpret_cast<char *>(smem);
    AxisSampleN *ytab = reinterpret_cast<AxisSampleN *>(smem + kT2Ch * ts + (4 - (kT2Ch * ts) % 4) % 4);
    AxisSampleN *xtab = ytab + kT2Axis;

    const int64_t r = blockIdx.x;
    const int c0 = blockIdx.y * kT2Ch;
    const float *roi = rois + r * 5;
    const int b = (int)roi[0];
    const float off = aligned ? 0.5f : 0.0f;
    const float start_w = roi[1] * scale - off, start_h = roi[2] * scale - off;
    const float end_w = roi[3] * scale - off, end_h = roi[4] * scale - off;
    float rw = end_w - start_w, rh = end_h - start_h;
    if (!aligned) {
        rw = fmaxf(rw, 1.f);
        rh = fmaxf(rh, 1.f);
    }
    const float bin_h = rh / (float)PH, bin_w = rw / (float)PW;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(bin_h);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(bin_w);
    const int prod = gh * gw;
    const float count = (float)(prod > 1 ? prod : 1);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int QN = kT2Ch / 4;                             // lanes (channel quads) per bin / per pixel
    constexpr int BPW = 64 / QN;                              // bins per wave instruction
    constexpr int PPP = kT2Threads / QN;                      // pixels (and bins) per pass of the workgroup
    const int q = lane % QN, sub = lane / QN;
    const int cq = c0 + 4 * q;
    const bool c_ok = cq < C;                                 // C % 4 == 0: a quad is all-in or all-out
    const unsigned ystride = (unsigned)W * C * (unsigned)sizeof(float), xstride = (unsigned)C * (unsigned)sizeof(float);
    const float *img = feat + (int64_t)b * H * W * C;
    const __amdgpu_buffer_rsrc_t img_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(img), 0, (unsigned)H * ystride, 0x00020000);

    // 1. the window's pixels, every load of this thread in flight: pixel p = (row p / ww, column p % ww), its 128 bytes by 8 lanes
    const int npx = wh * ww;
    const float inv_ww = 1.0f / (float)ww;
    const int p0 = threadIdx.x / QN, qq = threadIdx.x % QN;
    const unsigned qoff = (unsigned)(c0 + 4 * qq < C ? c0 + 4 * qq : 0) * (unsigned)sizeof(float);
    float4 stage[kWinMaxPasses];
#pragma unroll
    for (int i = 0; i < kWinMaxPasses; i++) {
        const int p = p0 + i * PPP;
        stage[i] = float4{0.f, 0.f, 0.f, 0.f};
        if (p < npx) {
            const int wr = (int)(((float)p + 0.5f) * inv_ww), wc = p - wr * ww;         // exact for these small integers
            stage[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      img_rsrc, (unsigned)(y0 + wr) * ystride + (unsigned)(x0 + wc) * xstride + qoff, 0, 0));
        }
    }
    // 2. (under those loads) the sampling tables, with WINDOW byte offsets: row offset for y, pixel offset for x.  A sample outside
    // [-1, size] has weights 0 and points at the window's first row / column (0 * finite = 0, as in the direct form)
    const int ny = PH * gh, nx = PW * gw;
    for (int t = threadIdx.x; t < ny; t += kT2Threads) {
        AxisSampleN a = axis_sample_n(start_h, bin_h, t / gh, t % gh, gh, H);
        a.lo = min(max(a.lo - y0, 0), wh - 1) * (ww * kWinPitch);
        a.hi = min(max(a.hi - y0, 0), wh - 1) * (ww * kWinPitch);
        ytab[t] = a;
    }
    for (int t = threadIdx.x; t < nx; t += kT2Threads) {
        AxisSampleN a = axis_sample_n(start_w, bin_w, t / gw, t % gw, gw, W);
        a.lo = min(max(a.lo - x0, 0), ww - 1) * kWinPitch;
        a.hi = min(max(a.hi - x0, 0), ww - 1) * kWinPitch;
        xtab[t] = a;
    }
#pragma unroll
    for (int i = 0; i < kWinMaxPasses; i++) {
        const int p = p0 + i * PPP;
        if (p < npx) *reinterpret_cast<float4 *>(win + p * kWinPitch + qq * 16) = stage[i];
    }
    __syncthreads();

    // 3. every bin of this thread out of the window, torchvision's sample order, un-fused: the arithmetic of the direct form
    const int ns = gh * gw;
    const float inv_pw = 1.0f / (float)PW;
    const int icount = prod > 1 ? prod : 1;
    const bool count_pow2 = (icount & (icount - 1)) == 0;     // wave-uniform
    const float inv_count = 1.0f / count;                     // exact when count is a power of two
    const char *wq = win + q * 16;
    float4 res[kWinMaxPasses];
#pragma unroll
    for (int k = 0; k < kWinMaxPasses; k++) {
        const int bin = k * PPP + wave * BPW + sub;
        const bool bin_ok = bin < bins;
        const int ph = bin_ok ? (int)(((float)bin + 0.5f) * inv_pw) : 0, pw = bin_ok ? bin - ph * PW : 0;
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        if (bin_ok && c_ok) {
            int iy = 0, ix = 0;
            for (int sidx = 0; sidx < ns; sidx++) {
                const AxisSampleN ys = ytab[ph * gh + iy], xs = xtab[pw * gw + ix];
                const float w1 = ys.wh * xs.wh, w2 = ys.wh * xs.wl, w3 = ys.wl * xs.wh, w4 = ys.wl * xs.wl;
                const float4 v1 = *reinterpret_cast<const float4 *>(wq + ys.lo + xs.lo);
                const float4 v2 = *reinterpret_cast<const float4 *>(wq + ys.lo + xs.hi);
                const float4 v3 = *reinterpret_cast<const float4 *>(wq + ys.hi + xs.lo);
                const float4 v4 = *reinterpret_cast<const float4 *>(wq + ys.hi + xs.hi);
                // ((w1*v1 + w2*v2) + w3*v3) + w4*v4, then accumulate -- un-fused (file built with -ffp-contract=off)
                acc.x = acc.x + (((w1 * v1.x + w2 * v2.x) + w3 * v3.x) + w4 * v4.x);
                acc.y = acc.y + (((w1 * v1.y + w2 * v2.y) + w3 * v3.y) + w4 * v4.y);
                acc.z = acc.z + (((w1 * v1.z + w2 * v2.z) + w3 * v3.z) + w4 * v4.z);
                acc.w = acc.w + (((w1 * v1.w + w2 * v2.w) + w3 * v3.w) + w4 * v4.w);
                if (++ix == gw) {
                    ix = 0;
                    iy++;
                }
            }
        }
        if (count_pow2) {              // x / 2^k == x * 2^-k bit for bit (both are the correctly rounded quotient)
            acc.x *= inv_count; acc.y *= inv_count; acc.z *= inv_count; acc.w *= inv_count;
        } else {
            acc.x /= count; acc.y /= count; acc.z /= count; acc.w /= count;
        }
        res[k] = acc;
    }
    __syncthreads();                                          // every tap has been read: the window's bytes become the tile
#pragma unroll
    for (int k = 0; k < kWinMaxPasses; k++) {
        const int bin = k * PPP + wave * BPW + sub;
        if (bin < bins) {
            float *t = tile + (4 * q) * ts + bin;
            t[0] = res[k].x;
            t[ts] = res[k].y;
            t[2 * ts] = res[k].z;
            t[3 * ts] = res[k].w;
        }
    }
    __syncthreads();
    const int cn = min(kT2Ch, C - c0);
    float *dst = out + (r * C + c0) * (int64_t)bins;
    if ((bins & 3) == 0) {
        const int qpc = bins >> 2;                                     // quads per channel
        int c = 0, b4 = threadIdx.x;
        while (b4 >= qpc) {
            b4 -= qpc;
            c++;
        }
        const int step_c = kT2Threads / qpc, step_b = kT2Threads - step_c * qpc;
        while (c < cn) {
            const float *t = tile + c * ts + 4 * b4;
            const float4 v = {t[0], t[1], t[2], t[3]};
            store4_policy<LOCOV_T2_STORE_AUX>(dst + (c * bins + 4 * b4), v);
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
    for (int idx = threadIdx.x; idx < cn * bins; idx += kT2Threads) {
        const int c = (int)(((float)idx + 0.5f) * inv_bins);       // idx / bins, exact for these sizes (no integer divide)
        dst[idx] = tile[c * ts + (idx - c * bins)];
    }
}

#ifndef LOCOV_T2_MINW
#define LOCOV_T2_MINW 4                                    // four waves per SIMD = four workgroups per CU (the register allocator's budget: 128)
#endif
__global__ __launch_bounds__(kT2Threads, LOCOV_T2_MINW) void roi_align_nhwc2nchw_kernel(
    const float *__restrict__ feat, int N, int H, int W, int C, const float *__restrict__ rois, int PH, int PW,
    float scale, int sampling_ratio, int aligned, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bins = PH * PW;
    const int ts = bins | 1;                                  // odd row stride of the transpose tile
    float *tile = smem;                                       // [kT2Ch][ts]
    AxisSampleN *ytab = reinterpret_cast<AxisSampleN *>(smem + kT2Ch * ts + (4 - (kT2Ch * ts) % 4) % 4);
    AxisSampleN *xtab = ytab + kT2Axis;

    const int64_t r = blockIdx.x;
    const int c0 = blockIdx.y * kT2Ch;
    const float *roi = rois + r * 5;
    const int b = (int)roi[0];
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
    const bool valid_b = b >= 0 && b < N;
    {
        // a proposal whose pixel rectangle fits the LDS window takes every tap from there (roi_align_window_path)
        int wy0, wx0, wwh, www;
        if (roi_window_rect(start_h, start_w, bin_h, bin_w, gh, gw, PH, PW, H, W, (kT2Ch * (bins | 1) * 4) / kWinPitch, valid_b, wy0, wx0, wwh, www)) {
            roi_align_window_path(feat, N, H, W, C, rois, PH, PW, scale, sampling_ratio, aligned, out, smem, wy0, wx0, wwh, www);
            return;
        }
    }
    gh = (gh > 0 && valid_b) ? gh : 0;
    gw = (gw > 0 && valid_b) ? gw : 0;
    const int ny = PH * gh, nx = PW * gw;
    const bool use_lds = ny <= kT2Axis && nx <= kT2Axis;
    // the tables hold BYTE offsets into the image (row offset for y, pixel offset for x): a tap address is
    // then two 32-bit adds on top of a wave-uniform buffer descriptor instead of 64-bit multiplies per tap
    const unsigned ystride = (unsigned)W * C * (unsigned)sizeof(float), xstride = (unsigned)C * (unsigned)sizeof(float);
    auto as_offsets = [](AxisSampleN a, unsigned stride) {
        a.lo = (int)((unsigned)a.lo * stride);
        a.hi = (int)((unsigned)a.hi * stride);
        return a;
    };
    if (use_lds) {
        for (int t = threadIdx.x; t < ny; t += kT2Threads)
            ytab[t] = as_offsets(axis_sample_n(start_h, bin_h, t / gh, t % gh, gh, H), ystride);
        for (int t = threadIdx.x; t < nx; t += kT2Threads)
            xtab[t] = as_offsets(axis_sample_n(start_w, bin_w, t / gw, t % gw, gw, W), xstride);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int QN = kT2Ch / 4;                             // lanes (channel quads) per bin
    constexpr int BPW = 64 / QN;                              // bins per wave instruction
    const int q = lane % QN, sub = lane / QN;
    const int cq = c0 + 4 * q;
    const bool c_ok = cq < C;                                 // C % 4 == 0: a quad is all-in or all-out
    const float *img = feat + (int64_t)(valid_b ? b : 0) * H * W * C;
    const __amdgpu_buffer_rsrc_t img_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(img), 0, (unsigned)H * ystride, 0x00020000);
    const unsigned ch_off = (unsigned)(c_ok ? cq : 0) * (unsigned)sizeof(float);
    auto tap = [&](unsigned off) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(img_rsrc, off, 0, 0));
    };
    const int ns = gh * gw;                                   // samples per bin
#ifndef LOCOV_T2_U
#define LOCOV_T2_U 2                                       // (4 until the row form took the 2-4-sample rows: 2 leaves it the registers -- mix 2.62 -> 2.50 ms)
#endif
    constexpr int U = LOCOV_T2_U;                             // samples in flight per lane on the plain path (4 x 16-byte loads each)
    const float inv_pw = 1.0f / (float)PW;
    const int icount = prod > 1 ? prod : 1;
    const bool count_pow2 = (icount & (icount - 1)) == 0;     // wave-uniform
    const float inv_count = 1.0f / count;                     // exact when count is a power of two
    for (int g0 = 0; g0 < bins; g0 += 4 * BPW) {
        const int bin = g0 + wave * BPW + sub;
        const bool bin_ok = bin < bins;
        // bin -> (ph, pw): exact for these small integers, and far cheaper than an integer division
        const int ph = bin_ok ? (int)(((float)bin + 0.5f) * inv_pw) : 0, pw = bin_ok ? bin - ph * PW : 0;
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        if (bin_ok && c_ok) {
            // The gather is latency-bound (taps come from L1 / L2), so the loads of up to U samples are
            // issued back to back before any of them is consumed; the accumulation still runs in sample
            // order (iy outer, ix inner), i.e. the oracle's order.  Groups are sized exactly (ns is
            // uniform per ROI): most ROIs have 1-4 samples per bin and padded groups would spend the
            // texture-address unit and the vector ALU -- both ~80 % busy here -- on duplicates.
            int iy = 0, ix = 0;                                  // sample counters (wave-uniform: scalar registers)
            auto group = [&](auto nu_tag) __attribute__((always_inline)) {
                constexpr int NU = decltype(nu_tag)::value;
                float4 v[NU][4];
                float w[NU][4];
#pragma unroll
                for (int u = 0; u < NU; u++) {
                    const AxisSampleN ys = use_lds ? ytab[ph * gh + iy]
                                                   : as_offsets(axis_sample_n(start_h, bin_h, ph, iy, gh, H), ystride);
                    const AxisSampleN xs = use_lds ? xtab[pw * gw + ix]
                                                   : as_offsets(axis_sample_n(start_w, bin_w, pw, ix, gw, W), xstride);
                    const unsigned xlo = (unsigned)xs.lo + ch_off, xhi = (unsigned)xs.hi + ch_off;
                    w[u][0] = ys.wh * xs.wh; w[u][1] = ys.wh * xs.wl; w[u][2] = ys.wl * xs.wh; w[u][3] = ys.wl * xs.wl;
                    v[u][0] = tap((unsigned)ys.lo + xlo);
                    v[u][1] = tap((unsigned)ys.lo + xhi);
                    v[u][2] = tap((unsigned)ys.hi + xlo);
                    v[u][3] = tap((unsigned)ys.hi + xhi);
                    if (++ix == gw) {
                        ix = 0;
                        iy++;
                    }
                }
#pragma unroll
                for (int u = 0; u < NU; u++) {
                    // ((w1*v1 + w2*v2) + w3*v3) + w4*v4, then accumulate -- un-fused (file built
                    // with -ffp-contract=off)
                    acc.x = acc.x + (((w[u][0] * v[u][0].x + w[u][1] * v[u][1].x) + w[u][2] * v[u][2].x) + w[u][3] * v[u][3].x);
                    acc.y = acc.y + (((w[u][0] * v[u][0].y + w[u][1] * v[u][1].y) + w[u][2] * v[u][2].y) + w[u][3] * v[u][3].y);
                    acc.z = acc.z + (((w[u][0] * v[u][0].z + w[u][1] * v[u][1].z) + w[u][2] * v[u][2].z) + w[u][3] * v[u][3].z);
                    acc.w = acc.w + (((w[u][0] * v[u][0].w + w[u][1] * v[u][1].w) + w[u][2] * v[u][2].w) + w[u][3] * v[u][3].w);
                }
            };
            // Rows of 2-4 samples: consecutive samples of a bin row are at most one pixel apart (grid = ceil(bin size)), so sample
            // ix + 1 re-uses one of sample ix's two pixel columns -- its left column IS the previous left or the previous right one.
            // A row of GW samples then needs GW + 1 columns x 2 rows of loads instead of 4 GW taps (6 / 8 / 10 instead of 8 / 12 / 16):
            // a quarter to three eighths fewer bytes through the texture path, which is what bounds the large proposals.  Which column is
            // re-used differs per lane group (= per bin): a select on already loaded registers, the SAME values in the same sample
            // order -- bit-identical.  Checked per row from the tables alone (all lanes must be able to re-use); otherwise the row
            // takes the four taps per sample.
            auto row_group = [&](auto gw_tag, auto rp_tag) __attribute__((always_inline)) {
                constexpr int GW = decltype(gw_tag)::value, RP = decltype(rp_tag)::value;
                AxisSampleN xs[GW];
                bool ok = true;
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    xs[i] = xtab[pw * gw + i];
                    if (i > 0) ok = ok && (xs[i].lo == xs[i - 1].lo || xs[i].lo == xs[i - 1].hi);
                }
                if (!__all(ok)) {                              // (wave-uniform: a lane group whose samples jump takes the plain form with it)
                    for (int t = 0; t < RP * GW; t++) group(std::integral_constant<int, 1>{});
                    return;
                }
                float4 L[RP][GW + 1], Hh[RP][GW + 1];
                AxisSampleN ys[RP];
#pragma unroll
                for (int r = 0; r < RP; r++) {
                    ys[r] = ytab[ph * gh + iy + r];
                    const unsigned ylo = (unsigned)ys[r].lo + ch_off, yhi = (unsigned)ys[r].hi + ch_off;
                    L[r][0] = tap(ylo + (unsigned)xs[0].lo);
                    Hh[r][0] = tap(yhi + (unsigned)xs[0].lo);
#pragma unroll
                    for (int i = 0; i < GW; i++) {
                        L[r][i + 1] = tap(ylo + (unsigned)xs[i].hi);
                        Hh[r][i + 1] = tap(yhi + (unsigned)xs[i].hi);
                    }
                }
#pragma unroll
                for (int r = 0; r < RP; r++) {
                    float4 a0 = L[r][0], a2 = Hh[r][0], a1 = L[r][1], a3 = Hh[r][1];
#pragma unroll
                    for (int i = 0; i < GW; i++) {
                        if (i > 0) {
                            const bool same = xs[i].lo == xs[i - 1].lo;      // else the previous right column
                            a0.x = same ? a0.x : a1.x; a0.y = same ? a0.y : a1.y; a0.z = same ? a0.z : a1.z; a0.w = same ? a0.w : a1.w;
                            a2.x = same ? a2.x : a3.x; a2.y = same ? a2.y : a3.y; a2.z = same ? a2.z : a3.z; a2.w = same ? a2.w : a3.w;
                            a1 = L[r][i + 1];
                            a3 = Hh[r][i + 1];
                        }
                        const float w0 = ys[r].wh * xs[i].wh, w1 = ys[r].wh * xs[i].wl, w2 = ys[r].wl * xs[i].wh, w3 = ys[r].wl * xs[i].wl;
                        acc.x = acc.x + (((w0 * a0.x + w1 * a1.x) + w2 * a2.x) + w3 * a3.x);
                        acc.y = acc.y + (((w0 * a0.y + w1 * a1.y) + w2 * a2.y) + w3 * a3.y);
                        acc.z = acc.z + (((w0 * a0.z + w1 * a1.z) + w2 * a2.z) + w3 * a3.z);
                        acc.w = acc.w + (((w0 * a0.w + w1 * a1.w) + w2 * a2.w) + w3 * a3.w);
                    }
                }
                iy += RP;                                      // (ix stays 0: whole rows)
            };
            // (Sharing a pixel ROW between two consecutive sample rows the same way -- 3 (GW + 1) loads per row pair instead of
            //  4 (GW + 1) -- was built too: bit-identical and 15-40 % SLOWER, 22 spilled registers at four waves per SIMD; R4.8.)
#ifndef LOCOV_T2_DEDUPE
#define LOCOV_T2_DEDUPE 1                                  // developer A/B: 0 = four taps per sample everywhere
#endif
            using std::integral_constant;
            if (LOCOV_T2_DEDUPE && use_lds && gw >= 2 && gw <= 4) {
                while (iy < gh) {
                    // (two rows in flight only at GW = 2: 12 loads; three samples x two rows asked for 140 registers = a wave per SIMD less)
                    if (gw == 2) {
                        if (iy + 1 < gh) row_group(integral_constant<int, 2>{}, integral_constant<int, 2>{});
                        else row_group(integral_constant<int, 2>{}, integral_constant<int, 1>{});
                    } else if (gw == 3) {
                        row_group(integral_constant<int, 3>{}, integral_constant<int, 1>{});
                    } else {
                        row_group(integral_constant<int, 4>{}, integral_constant<int, 1>{});
                    }
                }
            } else {
                int s0 = 0;
                for (; s0 + U <= ns; s0 += U) group(std::integral_constant<int, U>{});
                switch (ns - s0) {                                   // wave-uniform remainder, 0..U-1 samples
                case 3: group(std::integral_constant<int, (U > 3 ? 3 : 1)>{}); break;
                case 2: group(std::integral_constant<int, (U > 2 ? 2 : 1)>{}); break;
                case 1: group(std::integral_constant<int, 1>{}); break;
                default: break;
                }
            }
        }
        if (bin_ok) {
            float *t = tile + (4 * q) * ts + bin;
            if (count_pow2) {          // x / 2^k == x * 2^-k bit for bit (both are the correctly rounded quotient)
                t[0] = acc.x * inv_count;
                t[ts] = acc.y * inv_count;
                t[2 * ts] = acc.z * inv_count;
                t[3 * ts] = acc.w * inv_count;
            } else {
                t[0] = acc.x / count;
                t[ts] = acc.y / count;
                t[2 * ts] = acc.z / count;
                t[3 * ts] = acc.w / count;
            }
        }
    }
    __syncthreads();
    const int cn = min(kT2Ch, C - c0);
    float *dst = out + (r * C + c0) * (int64_t)bins;
    if ((bins & 3) == 0) {
        // 16 bytes per lane: four consecutive bins of one channel (a channel's run is a multiple of 4 floats, so a quad never
        // straddles two channels); (channel, bin quad) advance incrementally -- no division per element
        const int qpc = bins >> 2;                                     // quads per channel
        int c = 0, b4 = threadIdx.x;
        while (b4 >= qpc) {
            b4 -= qpc;
            c++;
        }
        const int step_c = kT2Threads / qpc, step_b = kT2Threads - step_c * qpc;
        while (c < cn) {
            const float *t = tile + c * ts + 4 * b4;
            const float4 v = {t[0], t[1], t[2], t[3]};
            store4_policy<LOCOV_T2_STORE_AUX>(dst + (c * bins + 4 * b4), v);
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
    for (int idx = threadIdx.x; idx < cn * bins; idx += kT2Threads) {
        const int c = (int)(((float)idx + 0.5f) * inv_bins);       // idx / bins, exact for these sizes (no integer divide)
        dst[idx] = tile[c * ts + (idx - c * bins)];
    }
}

// channel slices per ROI of roi_align_nhwc_kernel: a power of two up to 8 (= one per XCD, see the kernel) that still leaves a
// slice at least 64 channel quads wide, so that a wave stays inside one bin (wave-uniform pixel loops): 2 048+ channels -> 8,
// 1 024 -> 4, 512 -> 2, fewer -> 1
// roi_align_tiles.hip: the LDS-staged form of the contract (mode LOCOV_ROIALIGN_FAST)
int64_t roi_align_tiles_plan_bytes(int64_t R);
int launch_roi_align_tiles(const float *feat_nhwc, int N, int H, int W, int C, const float *rois, int64_t R, int PH, int PW,
                           float scale, int sampling_ratio, int aligned, void *plan_ws, float *out, hipStream_t s);

// ---- the even-grid pooler's backward by OWNERSHIP: one workgroup = one 8 x 8 pixel tile of one image x one 128-channel slice ---------
//
// roi_align_nhwc_kernel<BWD> gives every (proposal, slice) a workgroup and adds each bin's contributions to the map with fp32
// memory-side atomics: 0.9-1.5 T atomic lanes per second, four times the scattered rate of the units (docs/experiments.md R5.23), and
// still 0.64 / 1.15 ms of the LSM / STT step for 0.16 / 0.3 GB of gradient rows.  Here the map is cut into tiles and a workgroup
// COLLECTS: it lists (in proposal order) the proposals of its image whose footprint reaches its tile, and for each of them every wave
// builds the separable per-pixel weights of the seven bin rows / columns on ITS eight pixel rows / four pixel columns (the sums of the
// samples' bilinear weights, as in the forward's separable form; no barrier between the waves inside the list), reads the gradient
// rows of the bins that reach the tile (its 128 channels: 512 contiguous bytes per bin) and adds  sum_oh w_y[oh][py] (sum_ow w_x[ow][px] g[oh][ow])  to REGISTER accumulators: a thread owns one
// channel and the 8 x 4 pixels of its column parity (two small dense products per proposal, at most 420 FMAs, instead of sparse updates).
// The tile is written (added to what the map gradient already holds) once: no atomics, and a sum whose order is the proposals'
// order -- the result is reproducible bit for bit.
constexpr int kBT = 8, kBwdCh = 128, kBwdList = 2048;

__global__ __launch_bounds__(256, 3) void roi_align_even_bwd_tiles_kernel(const float *__restrict__ grad_rows, int64_t grad_ld,
                                                                      const float *__restrict__ rois, int R, int N, int H, int W, int C, int PH,
                                                                      int PW, float scale, int sampling_ratio, int aligned, int bin_stride,
                                                                      float *__restrict__ grad_feat, int nslices, int tiles_x, int tiles_y)
{
    constexpr int OB = 7;
    __shared__ float wy[4][OB][kBT];                              // per WAVE: weights of the bin rows on the tile's pixel rows
    __shared__ float4 wx[4][OB];                                  // per wave: weights of the bin columns on the four tile columns of its parity
    __shared__ unsigned short list[kBwdList];
    __shared__ int wave_cnt[4], list_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int slice = t % nslices;                                // (= the XCD under round-robin dispatch: an XCD reads ONE channel slice)
    t /= nslices;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, img = t / tiles_y;
    const int ty0 = ty * kBT, tx0 = tx * kBT, c0 = slice * kBwdCh;
    const float off = aligned ? 0.5f : 0.0f;
    // this thread's accumulators: channel c, the 8 pixel rows x the 4 pixel columns of its parity
    float acc_r[kBT][kBT / 2];
#pragma unroll
    for (int py = 0; py < kBT; py++)
#pragma unroll
        for (int q = 0; q < kBT / 2; q++) acc_r[py][q] = 0.f;

    struct Geo { float start_w, start_h, bin_w, bin_h; int gw, gh; };
    auto geometry = [&](const float *roi) {
        Geo g;
        g.start_w = __fsub_rn(__fmul_rn(roi[1], scale), off);
        g.start_h = __fsub_rn(__fmul_rn(roi[2], scale), off);
        const float end_w = __fsub_rn(__fmul_rn(roi[3], scale), off), end_h = __fsub_rn(__fmul_rn(roi[4], scale), off);
        float rw = __fsub_rn(end_w, g.start_w), rh = __fsub_rn(end_h, g.start_h);
        if (!aligned) {
            rw = fmaxf(rw, 1.f);
            rh = fmaxf(rh, 1.f);
        }
        g.bin_h = __fdiv_rn(rh, (float)PH);
        g.bin_w = __fdiv_rn(rw, (float)PW);
        g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(g.bin_h);
        g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(g.bin_w);
        return g;
    };
    const int c = tid & (kBwdCh - 1), half = tid >> 7;

    for (int base = 0; base < R; base += kBwdList) {
        // ---- the proposals of this image whose footprint (conservatively: the box in map pixels, two pixels wider) reaches the tile,
        //      in proposal order (ballots + prefix counts: the order of the sums below must not depend on timing)
        if (tid == 0) list_n = 0;
        __syncthreads();
        const int stop = min(R, base + kBwdList);
        for (int r0 = base; r0 < stop; r0 += 256) {
            const int r = r0 + tid;
            bool ok = false;
            if (r < stop) {
                const float *roi = rois + (int64_t)r * 5;
                if ((int)roi[0] == img) {
                    const Geo g = geometry(roi);
                    const float y_lo = g.start_h - 2.f, y_hi = g.start_h + g.bin_h * (float)PH + 2.f;
                    const float x_lo = g.start_w - 2.f, x_hi = g.start_w + g.bin_w * (float)PW + 2.f;
                    // (NaN coordinates fail every comparison: such a proposal contributes nothing here, as its samples are invalid there)
                    ok = y_hi >= (float)ty0 && y_lo <= (float)(ty0 + kBT) && x_hi >= (float)tx0 && x_lo <= (float)(tx0 + kBT) && g.gh > 0 && g.gw > 0;
                }
            }
            const unsigned long long b = __ballot(ok);
            if (lane == 0) wave_cnt[wave] = __popcll(b);
            __syncthreads();
            int pos = list_n;
            for (int w = 0; w < wave; w++) pos += wave_cnt[w];
            if (ok) list[pos + __popcll(b & ((1ull << lane) - 1ull))] = (unsigned short)(r - base);
            __syncthreads();
            if (tid == 0) list_n += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            __syncthreads();
        }
        const int n_list = list_n;
        // Every WAVE builds the tables it uses (lanes 0-55: the rows, then the columns of its parity) in its own corner of LDS: the
        // four waves of the workgroup never wait for each other inside the list.  my / mx: bit 8 o + p set when bin row / column o has
        // a weight on tile row / column p (ballots: they stay in scalar registers).
        float (*wyw)[kBT] = wy[wave];
        float4 *wxw = wx[wave];
        for (int li = 0; li < n_list; li++) {
            const int r = base + (int)list[li];
            const Geo g = geometry(rois + (int64_t)r * 5);
            unsigned long long my, mx;
            {
                const int o = lane / kBT, p = lane % kBT;
                float w = 0.f;
                if (lane < OB * kBT)
                    for (int i = 0; i < g.gh; i++) {
                        const AxisSampleN sm = axis_sample_n(g.start_h, g.bin_h, o * bin_stride, i, g.gh, H);
                        w += (sm.lo == ty0 + p ? sm.wh : 0.f) + (sm.hi == ty0 + p ? sm.wl : 0.f);
                    }
                my = __ballot(w != 0.f);
                if (lane < OB * kBT) wyw[o][p] = w;
                w = 0.f;
                if (lane < OB * kBT)
                    for (int i = 0; i < g.gw; i++) {
                        const AxisSampleN sm = axis_sample_n(g.start_w, g.bin_w, o * bin_stride, i, g.gw, W);
                        w += (sm.lo == tx0 + p ? sm.wh : 0.f) + (sm.hi == tx0 + p ? sm.wl : 0.f);
                    }
                mx = __ballot(w != 0.f);
                if (lane < OB * kBT && (p & 1) == half) reinterpret_cast<float *>(&wxw[o])[p >> 1] = w;
            }
            // (the tables are read by OTHER lanes of this wave: order the LDS writes above and the reads below for the wave)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (my == 0 || mx == 0) continue;                      // (the conservative box reached the tile, no sample did)
            // ---- the gradient rows of the bins that reach the tile: all requested before any is used
            const int prod = g.gh * g.gw;
            const float inv_count = 1.f / (float)(prod > 1 ? prod : 1);
            const float *grow = grad_rows + (int64_t)r * (OB * OB) * grad_ld + c0 + c;
            float gv[OB * OB];
#pragma unroll
            for (int oh = 0; oh < OB; oh++)
#pragma unroll
                for (int ow = 0; ow < OB; ow++)
                    gv[oh * OB + ow] = ((my >> (8 * oh)) & 0xffull) != 0 && ((mx >> (8 * ow)) & 0xffull) != 0 ? grow[(int64_t)(oh * OB + ow) * grad_ld] : 0.f;
            // ---- in registers:  acc[py][px] += sum_oh wy[oh][py] * (sum_ow wx[ow][px] * g[oh][ow])   for this thread's 8 x 4 pixels
#pragma unroll
            for (int oh = 0; oh < OB; oh++) {
                if (((my >> (8 * oh)) & 0xffull) == 0) continue;     // (wave-uniform: a bin row without a weight on this tile)
                float tq[kBT / 2] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ow = 0; ow < OB; ow++) {
                    const float4 w4 = wxw[ow];                        // this wave's four columns (its parity) of the column weights
                    const float gq = gv[oh * OB + ow];
                    tq[0] = fmaf(w4.x, gq, tq[0]);
                    tq[1] = fmaf(w4.y, gq, tq[1]);
                    tq[2] = fmaf(w4.z, gq, tq[2]);
                    tq[3] = fmaf(w4.w, gq, tq[3]);
                }
#pragma unroll
                for (int q = 0; q < kBT / 2; q++) tq[q] *= inv_count;
#pragma unroll
                for (int py = 0; py < kBT; py++) {
                    const float w = wyw[oh][py];
#pragma unroll
                    for (int q = 0; q < kBT / 2; q++) acc_r[py][q] = fmaf(w, tq[q], acc_r[py][q]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the next proposal's tables overwrite these)
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();                                            // (the list is rebuilt by the next pass)
    }
    __syncthreads();
    // ---- the tile, added to what the map gradient holds (a wave writes 256 contiguous bytes per pixel)
#pragma unroll
    for (int py = 0; py < kBT; py++)
#pragma unroll
        for (int q = 0; q < kBT / 2; q++) {
            const int y = ty0 + py, x = tx0 + 2 * q + half;
            const float v = acc_r[py][q];
            if (y < H && x < W && v != 0.f) grad_feat[(((int64_t)img * H + y) * W + x) * C + c0 + c] += v;
        }
}

static int nhwc_slices(int C)
{
    static const int forced = [] { const char *e = getenv("LOCOV_ROIALIGN_SLICES"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced;
    int n = 1;
    while (n < 8 && (C >> 2) / (2 * n) >= 64) n *= 2;
    return n;
}

// ROIAlign (even bins of a 14 x 14 pooler = 7 x 7, ROI-major) + per-channel affine + ReLU + Winograd input transform in one launch
bool roi_align_nhwc_wino_applicable(int C, int64_t R)
{
    const char *fe = getenv("LOCOV_WINO_FUSE");            // developer A/B / tests (read per launch): 0 = never
    if (fe && atoi(fe) == 0) return false;
    return C % 64 == 0 && R * (C / 64) <= 0x7fffffffLL;
}

int launch_roi_align_nhwc_wino(const float *feat, int N, int H, int W, int C, int64_t feat_ld, const float *rois, int64_t R, int pooled,
                               float spatial_scale, int sampling_ratio, int aligned, const float *ch_scale, const float *ch_shift, int relu,
                               float *V, float v_scale, unsigned *overflow, hipStream_t s)
{
    // (LOCOV_WINO_SLICE_CH: developer A/B of the slice width)
    static const int forced = [] { const char *e = getenv("LOCOV_WINO_SLICE_CH"); return e ? atoi(e) : 0; }();
    const int width = forced == 64 || forced == 128 ? (C % forced == 0 ? forced : 64) : (C % 128 == 0 ? 128 : 64);
    const int nslices = C / width;
    if (width == 128)
        hipLaunchKernelGGL((roi_align_nhwc_kernel<float, float, false, 128>), dim3((unsigned)(R * nslices)), dim3(kNhwcThreads), 0, s, feat, N, H, W, C,
                           rois, pooled, pooled, spatial_scale, sampling_ratio, aligned, 2, (pooled + 1) / 2, (pooled + 1) / 2, 0, V, (int64_t)C, feat_ld,
                           ch_scale, ch_shift, relu, nslices, R, v_scale, overflow);
    else
        hipLaunchKernelGGL((roi_align_nhwc_kernel<float, float, false, 64>), dim3((unsigned)(R * nslices)), dim3(kNhwcThreads), 0, s, feat, N, H, W, C,
                           rois, pooled, pooled, spatial_scale, sampling_ratio, aligned, 2, (pooled + 1) / 2, (pooled + 1) / 2, 0, V, (int64_t)C, feat_ld,
                           ch_scale, ch_shift, relu, nslices, R, v_scale, overflow);
    return check_launch("locov_roi_align_winograd_conv3x3_f32_split (ROIAlign + input transform)");
}

}  // namespace locov

using namespace locov;

extern "C" {

int64_t locov_roi_align_plan_bytes(int64_t R) { return R > 0 ? roi_align_tiles_plan_bytes(R) : 0; }

int locov_roi_align_from_nhwc_fwd_ex(const float *feat_nhwc, int N, int H, int W, int C, const float *rois, int64_t R,
                                     int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                                     int mode, void *workspace, int64_t workspace_bytes, float *out, locov_stream_t stream)
{
    LOCOV_REQUIRE(mode == LOCOV_ROIALIGN_EXACT || mode == LOCOV_ROIALIGN_FAST, "locov_roi_align_from_nhwc_fwd: bad mode %d", mode);
    LOCOV_REQUIRE(R >= 0, "locov_roi_align_from_nhwc_fwd: R < 0");
    LOCOV_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "locov_roi_align_from_nhwc_fwd: bad feature shape");
    LOCOV_REQUIRE(pooled_h > 0 && pooled_w > 0, "locov_roi_align_from_nhwc_fwd: bad pooled size");
    LOCOV_REQUIRE(spatial_scale > 0.f, "locov_roi_align_from_nhwc_fwd: spatial_scale must be > 0");
    LOCOV_REQUIRE(C % 4 == 0, "locov_roi_align_from_nhwc_fwd: C must be a multiple of 4");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(feat_nhwc && rois && out, "locov_roi_align_from_nhwc_fwd: null pointer");
    LOCOV_REQUIRE(R <= 0x7fffffffLL, "locov_roi_align_from_nhwc_fwd: R too large");
    LOCOV_REQUIRE((int64_t)H * W * C * 4 < 0xffffffffLL, "locov_roi_align_from_nhwc_fwd: one image must stay below 4 GiB");
    if (mode == LOCOV_ROIALIGN_FAST) {
        LOCOV_REQUIRE(workspace && workspace_bytes >= locov_roi_align_plan_bytes(R) && (uintptr_t)workspace % 16 == 0,
                      "locov_roi_align_from_nhwc_fwd: the fast form needs locov_roi_align_plan_bytes(R) bytes of 16-byte aligned workspace");
        return launch_roi_align_tiles(feat_nhwc, N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, workspace,
                                      out, as_stream(stream));
    }
    const int bins = pooled_h * pooled_w, ts = bins | 1;
    const size_t lds = ((size_t)kT2Ch * ts + 4) * sizeof(float) + 2 * kT2Axis * sizeof(AxisSampleN);
    LOCOV_REQUIRE(lds <= 150 * 1024, "locov_roi_align_from_nhwc_fwd: pooled size %dx%d too large for the LDS tile", pooled_h,
                  pooled_w);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(roi_align_nhwc2nchw_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return set_error(LOCOV_ERR_LAUNCH, "locov_roi_align_from_nhwc_fwd: cannot raise the dynamic LDS limit to %zu bytes", (size_t)lds);
    dim3 grid((unsigned)R, (unsigned)ceil_div(C, kT2Ch));
    hipLaunchKernelGGL(roi_align_nhwc2nchw_kernel, grid, dim3(kT2Threads), lds, as_stream(stream), feat_nhwc, N, H, W, C,
                       rois, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, out);
    return check_launch("locov_roi_align_from_nhwc_fwd");
}

int locov_roi_align_from_nhwc_fwd(const float *feat_nhwc, int N, int H, int W, int C, const float *rois, int64_t R,
                                  int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                                  float *out, locov_stream_t stream)
{
    return locov_roi_align_from_nhwc_fwd_ex(feat_nhwc, N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale, sampling_ratio,
                                            aligned, LOCOV_ROIALIGN_EXACT, nullptr, 0, out, stream);
}

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
    return locov_roi_align_nhwc_ld_fwd(feat, feat_dtype, N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale,
                                       sampling_ratio, aligned, bin_stride, pos_major, out, (int64_t)C, out_dtype, stream);
}

int locov_roi_align_nhwc_ld_fwd(const void *feat, int feat_dtype, int N, int H, int W, int C, const float *rois,
                                int64_t R, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                int aligned, int bin_stride, int pos_major, void *out, int64_t out_ld, int out_dtype,
                                locov_stream_t stream)
{
    return locov_roi_align_nhwc_affine_fwd(feat, feat_dtype, N, H, W, C, (int64_t)C, rois, R, pooled_h, pooled_w,
                                           spatial_scale, sampling_ratio, aligned, bin_stride, pos_major, nullptr, nullptr, 0,
                                           out, out_ld, out_dtype, stream);
}

int locov_roi_align_nhwc_bwd(const float *grad_rows, int64_t grad_ld, int N, int H, int W, int C, const float *rois, int64_t R,
                             int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int aligned, int bin_stride,
                             int pos_major, float *grad_feat, locov_stream_t stream)
{
    LOCOV_REQUIRE(grad_ld >= C && grad_ld % 4 == 0, "locov_roi_align_nhwc_bwd: grad_ld must be >= C and a multiple of 4");
    LOCOV_REQUIRE((int64_t)H * W * C * 4 < 0xffffffffLL, "locov_roi_align_nhwc_bwd: one image must stay below 4 GiB");
    LOCOV_REQUIRE(R >= 0 && N > 0 && C > 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0, "locov_roi_align_nhwc_bwd: bad shape");
    LOCOV_REQUIRE(spatial_scale > 0.f, "locov_roi_align_nhwc_bwd: spatial_scale must be > 0");
    LOCOV_REQUIRE(bin_stride == 1 || bin_stride == 2, "locov_roi_align_nhwc_bwd: bin_stride must be 1 or 2");
    LOCOV_REQUIRE(C % 4 == 0, "locov_roi_align_nhwc_bwd: C must be a multiple of 4");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(grad_rows && rois && grad_feat, "locov_roi_align_nhwc_bwd: null pointer");
    LOCOV_REQUIRE(R <= 0x7fffffffLL, "locov_roi_align_nhwc_bwd: R too large");
    LOCOV_REQUIRE(((uintptr_t)grad_rows | (uintptr_t)grad_feat) % 16 == 0, "locov_roi_align_nhwc_bwd: misaligned pointer");
    const int OH = (pooled_h + bin_stride - 1) / bin_stride, OW = (pooled_w + bin_stride - 1) / bin_stride;
    // the ownership form (7 x 7 bins, ROI-major rows, 128-channel slices): developer A/B LOCOV_POOL_BWD_TILES=0 -> the scatter below
    {
        const char *te = getenv("LOCOV_POOL_BWD_TILES");      // (read per launch: tests flip it)
        if ((!te || atoi(te) != 0) && OH == 7 && OW == 7 && !pos_major && C % kBwdCh == 0) {
            const int tiles_x = (W + kBT - 1) / kBT, tiles_y = (H + kBT - 1) / kBT, ns = C / kBwdCh;
            const int64_t wgs = (int64_t)N * tiles_x * tiles_y * ns;
            LOCOV_REQUIRE(wgs <= 0x7fffffffLL, "locov_roi_align_nhwc_bwd: map too large");
            hipLaunchKernelGGL(roi_align_even_bwd_tiles_kernel, dim3((unsigned)wgs), dim3(256), 0, as_stream(stream), grad_rows, grad_ld, rois, (int)R,
                               N, H, W, C, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, bin_stride, grad_feat, ns, tiles_x, tiles_y);
            return check_launch("locov_roi_align_nhwc_bwd (tiles)");
        }
    }
    // (slices of at most 128 channels where C allows: the 20 KB gradient window then holds the 40 pixels of a proposal below ~70 px)
    int nslices = nhwc_slices(C);
    while (nslices < 8 && C / (2 * nslices) >= 128 && (C >> 2) % (2 * nslices) == 0) nslices *= 2;
    LOCOV_REQUIRE(R * nslices <= 0x7fffffffLL, "locov_roi_align_nhwc_bwd: R too large");
    dim3 grid((unsigned)(R * nslices));
    // Window size: measured on the LSM step's launch (800 proposals, 4 images, 1024 channels) / the STT step's (1536 proposals, 3 images),
    // tools/ab_pool_bwd_sizes.py: none 0.771 / 1.437 ms, 16 KB 0.615 / 1.111, 20 KB 0.603 / 1.084, 24 KB 0.608 / 1.101, 32 KB 0.667 / 1.234,
    // 64 KB 0.835 -- a larger window takes more proposals but fewer workgroups per CU, and the window path needs the occupancy its two
    // barriers cost.  (Proposals of one small size class ALONE run slower through the window, 0.55 -> 0.71 ms: what it buys is room at the
    // memory-side atomic units for the large proposals' scatter, which bounds the launch.)
    // developer A/B: LOCOV_POOL_BWD_WINDOW=<bytes>, 0 -> every proposal scatters straight to memory
    const char *we = getenv("LOCOV_POOL_BWD_WINDOW");          // (read per launch: tests flip it)
    const int win_bytes = we ? atoi(we) : 20480;
    // (the attribute belongs to the kernel's code object on ONE device: set once per device a launch is made on -- a process that
    // drives several GPUs, or whose first call failed on one of them, must not decide for the others)
    static int attr_state[64] = {};                            // per device: 0 = not tried, 1 = set, -1 = refused
    int dev = 0;
    bool attr_ok = false;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (attr_state[dev] == 0)
            attr_state[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(&roi_align_nhwc_kernel<float, float, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 65536) == hipSuccess ? 1 : -1;
        attr_ok = attr_state[dev] == 1;
    }
    const int wb = attr_ok && win_bytes > 0 ? (win_bytes < 65536 ? win_bytes : 65536) : 0;
    hipLaunchKernelGGL((roi_align_nhwc_kernel<float, float, true>), grid, dim3(kNhwcThreads), (size_t)wb, as_stream(stream),
                       (const float *)grad_feat, N, H, W, C, rois, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, bin_stride,
                       OH, OW, pos_major, const_cast<float *>(grad_rows), grad_ld, (int64_t)C, (const float *)nullptr,
                       (const float *)nullptr, 0, nslices, R, 1.f, static_cast<unsigned *>(nullptr), wb / 4);
    return check_launch("locov_roi_align_nhwc_bwd");
}

int locov_roi_align_nhwc_affine_fwd(const void *feat, int feat_dtype, int N, int H, int W, int C, int64_t feat_ld,
                                    const float *rois, int64_t R, int pooled_h, int pooled_w, float spatial_scale,
                                    int sampling_ratio, int aligned, int bin_stride, int pos_major, const float *ch_scale,
                                    const float *ch_shift, int relu, void *out, int64_t out_ld, int out_dtype,
                                    locov_stream_t stream)
{
    LOCOV_REQUIRE(out_ld >= C && out_ld % 4 == 0, "locov_roi_align_nhwc_fwd: out_ld must be >= C and a multiple of 4");
    LOCOV_REQUIRE(feat_ld >= C && feat_ld % 4 == 0, "locov_roi_align_nhwc_fwd: feat_ld must be >= C and a multiple of 4");
    LOCOV_REQUIRE((int64_t)H * W * feat_ld * 4 < 0xffffffffLL, "locov_roi_align_nhwc_fwd: one image must stay below 4 GiB");
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
    const int nslices = nhwc_slices(C);
    LOCOV_REQUIRE(R * nslices <= 0x7fffffffLL, "locov_roi_align_nhwc_fwd: R too large");
    dim3 grid((unsigned)(R * nslices));
    hipStream_t s = as_stream(stream);
#define LOCOV_LAUNCH_NHWC(TI, TO)                                                                                   \
    hipLaunchKernelGGL((roi_align_nhwc_kernel<TI, TO>), grid, dim3(kNhwcThreads), 0, s, (const TI *)feat, N, H, W, C, \
                       rois, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, bin_stride, OH, OW, pos_major, (TO *)out, out_ld, \
                       feat_ld, ch_scale, ch_shift, relu, nslices, R)
    if (feat_dtype == LOCOV_F32 && out_dtype == LOCOV_F32) LOCOV_LAUNCH_NHWC(float, float);
    else if (feat_dtype == LOCOV_F32 && out_dtype == LOCOV_BF16) LOCOV_LAUNCH_NHWC(float, __bf16);
    else if (feat_dtype == LOCOV_BF16 && out_dtype == LOCOV_F32) LOCOV_LAUNCH_NHWC(__bf16, float);
    else LOCOV_LAUNCH_NHWC(__bf16, __bf16);
#undef LOCOV_LAUNCH_NHWC
    return check_launch("locov_roi_align_nhwc_fwd");
}

}  // extern "C"
