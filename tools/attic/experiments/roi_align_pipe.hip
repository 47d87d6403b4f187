// The pooler contract ([R,C,ph,pw] fp32 out of an NCHW map; roi_emb_heads.py:182-187,243-245) as a SOFTWARE-PIPELINED workgroup:
// the exact arithmetic of roi_align_nhwc2nchw_kernel (torchvision's per-sample order, un-fused -- bit-identical to the oracle), with
// the two phases that used to ADD now running beside each other.
//
// What was wrong with one workgroup per (ROI, 32 channels) (roi_align_nhwc.hip): every wave gathers (texture-address bound for the
// large boxes, latency bound for the small ones), then transposes through LDS, then stores 25 KB -- and a wave that has issued its
// stores cannot end, free its slot or start another gather until they retire.  Store-only runs at the HBM write roof (1.05 ms for
// 6.4 GB), gather-only takes 2.5 ms, together 3.0 ms (profiles/r03_roialign_contract_ablation.txt).
//
// Here one workgroup owns a ROI and walks its channel blocks (16 channels each) with TWO transpose tiles in LDS and waves with
// roles: four GATHER waves fill tile[i & 1] with block i (a tap = 64 contiguous bytes per bin, 16 bins per wave instruction)
// while one STORE wave drains tile[(i - 1) & 1] -- block i - 1 -- as 12.5 KB of contiguous NCHW output, 16 bytes per lane.  One
// workgroup barrier per block.  The gather waves never carry a store in their memory queue, the store wave's queue holds nothing
// else; the per-ROI sampling tables are built once per ROI instead of once per channel block.
//
// RESULT (round 4, 8 x 1000 bench proposals, 1024 channels; tools/ab_pipe.sh): bit-identical (the 27 ROIAlign tests pass) and
// 2x SLOWER -- 6.0 ms against 2.87 ms; with 3 / 2 gather waves 5.3 / 5.5 ms, at two workgroups per CU 8.4 ms.  The gather is a
// dependent chain per wave (tables -> 12-16 loads -> wait -> arithmetic -> tile), so its rate is (gather waves per CU) x (loads in
// flight per wave) / latency: the shipped kernel keeps 16 such waves per CU, this one 12 (a fifth of every workgroup's waves store,
// 106 VGPRs allow four waves per SIMD), loses a fifth of each block to the 4/3/3/3 split of its 13 wave groups behind a barrier per
// block, and fetches half cache lines (16 channels per block so that two tiles fit).  Overlapping the stores can return at most the
// 15 % they cost.  Parked: to be built as a library source again it needs `"roi_align_pipe.hip": ["-ffp-contract=off"]` in
// locov_amd/build.py and the two declarations + the call in locov_roi_align_from_nhwc_fwd_ex that round 4 removed.
#include "../../../locov_amd/csrc/roi_align_common.h"

#include <cstdlib>
#include <type_traits>

namespace locov {

namespace {

constexpr int kPipeCh = 16;                     // channels per block
#ifndef LOCOV_PIPE_GW
#define LOCOV_PIPE_GW 4
#endif
#ifndef LOCOV_PIPE_U
#define LOCOV_PIPE_U 3
#endif
#ifndef LOCOV_PIPE_MINW
#define LOCOV_PIPE_MINW 4
#endif
constexpr int kPipeGather = LOCOV_PIPE_GW;      // gather waves per workgroup (+ one store wave)
constexpr int kPipeThreads = 64 * (kPipeGather + 1);
constexpr int kPipeAxis = 128;                  // per-axis table entries (14 bins x up to 9 samples; larger grids: computed on the fly)
constexpr int kPipeQN = kPipeCh / 4;            // lanes (channel quads) per bin
constexpr int kPipeBPW = 64 / kPipeQN;          // bins per wave instruction

__device__ __forceinline__ void store4_nt(float *p, const float4 &v)
{
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const f32x4_t d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(d) : "memory");
}

}  // namespace

// grid (R, splits): workgroup (r, s) takes the channel blocks [s * blocks_per_wg, ...) of ROI r
__global__ __launch_bounds__(kPipeThreads, LOCOV_PIPE_MINW) void roi_align_pipe_kernel(const float *__restrict__ feat, int N, int H, int W, int C,
                                                                      const float *__restrict__ rois, int PH, int PW, float scale,
                                                                      int sampling_ratio, int aligned, float *__restrict__ out,
                                                                      int blocks_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bins = PH * PW;
    const int ts = bins | 1;                                  // odd row stride of a transpose tile
    const int tile_floats = (kPipeCh * ts + 3) & ~3;
    float *const tiles = smem;                                // [2][kPipeCh][ts]
    AxisSampleN *ytab = reinterpret_cast<AxisSampleN *>(smem + 2 * tile_floats);
    AxisSampleN *xtab = ytab + kPipeAxis;

    const int64_t r = blockIdx.x;
    const int nblk_all = (C + kPipeCh - 1) / kPipeCh;
    const int cb0 = blockIdx.y * blocks_per_wg;
    const int nb = min(blocks_per_wg, nblk_all - cb0);
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
    gh = (gh > 0 && valid_b) ? gh : 0;
    gw = (gw > 0 && valid_b) ? gw : 0;
    const int ny = PH * gh, nx = PW * gw;
    const bool use_lds = ny <= kPipeAxis && nx <= kPipeAxis;
    // tables of BYTE offsets into the image (row offset for y, pixel offset for x): a tap address is two 32-bit adds on top of a
    // wave-uniform buffer descriptor
    const unsigned ystride = (unsigned)W * C * (unsigned)sizeof(float), xstride = (unsigned)C * (unsigned)sizeof(float);
    auto as_offsets = [](AxisSampleN a, unsigned stride) {
        a.lo = (int)((unsigned)a.lo * stride);
        a.hi = (int)((unsigned)a.hi * stride);
        return a;
    };
    if (use_lds) {
        for (int t = threadIdx.x; t < ny; t += kPipeThreads)
            ytab[t] = as_offsets(axis_sample_n(start_h, bin_h, t / gh, t % gh, gh, H), ystride);
        for (int t = threadIdx.x; t < nx; t += kPipeThreads)
            xtab[t] = as_offsets(axis_sample_n(start_w, bin_w, t / gw, t % gw, gw, W), xstride);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *img = feat + (int64_t)(valid_b ? b : 0) * H * W * C;
    const __amdgpu_buffer_rsrc_t img_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(img), 0, (unsigned)H * ystride, 0x00020000);
    auto tap = [&](unsigned off) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(img_rsrc, off, 0, 0));
    };
    const int ns = gh * gw;                                   // samples per bin
    constexpr int U = LOCOV_PIPE_U;                           // samples in flight per lane (4 x 16-byte loads each): 3 -> 106 VGPRs, 4 waves per SIMD
    const float inv_pw = 1.0f / (float)PW;
    const int icount = prod > 1 ? prod : 1;
    const bool count_pow2 = (icount & (icount - 1)) == 0;     // wave-uniform
    const float inv_count = 1.0f / count;                     // exact when count is a power of two
    const int ngroups = (bins + kPipeBPW - 1) / kPipeBPW;
    const int q = lane % kPipeQN, sub = lane / kPipeQN;

    for (int i = 0; i <= nb; i++) {
        if (wave < kPipeGather) {
            if (i < nb) {
                // ---- gather block cb0 + i into tile[i & 1]: the arithmetic of roi_align_nhwc2nchw_kernel, bit for bit
                float *tile = tiles + (i & 1) * tile_floats;
                const int cq = (cb0 + i) * kPipeCh + 4 * q;
                const bool c_ok = cq < C;                     // C % 4 == 0: a quad is all-in or all-out
                const unsigned ch_off = (unsigned)(c_ok ? cq : 0) * (unsigned)sizeof(float);
                for (int g = wave; g < ngroups; g += kPipeGather) {
                    const int bin = g * kPipeBPW + sub;
                    const bool bin_ok = bin < bins;
                    // bin -> (ph, pw): exact for these small integers, and far cheaper than an integer division
                    const int ph = bin_ok ? (int)(((float)bin + 0.5f) * inv_pw) : 0, pw = bin_ok ? bin - ph * PW : 0;
                    float4 acc = {0.f, 0.f, 0.f, 0.f};
                    if (bin_ok && c_ok) {
                        int iy = 0, ix = 0;                   // sample counters (wave-uniform: scalar registers)
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
                                // ((w1*v1 + w2*v2) + w3*v3) + w4*v4, then accumulate -- un-fused (file built with -ffp-contract=off)
                                acc.x = acc.x + (((w[u][0] * v[u][0].x + w[u][1] * v[u][1].x) + w[u][2] * v[u][2].x) + w[u][3] * v[u][3].x);
                                acc.y = acc.y + (((w[u][0] * v[u][0].y + w[u][1] * v[u][1].y) + w[u][2] * v[u][2].y) + w[u][3] * v[u][3].y);
                                acc.z = acc.z + (((w[u][0] * v[u][0].z + w[u][1] * v[u][1].z) + w[u][2] * v[u][2].z) + w[u][3] * v[u][3].z);
                                acc.w = acc.w + (((w[u][0] * v[u][0].w + w[u][1] * v[u][1].w) + w[u][2] * v[u][2].w) + w[u][3] * v[u][3].w);
                            }
                        };
                        int s0 = 0;
                        for (; s0 + U <= ns; s0 += U) group(std::integral_constant<int, U>{});
                        switch (ns - s0) {                    // wave-uniform remainder, 0..U-1 samples
                        case 3: group(std::integral_constant<int, (U > 3 ? 3 : 1)>{}); break;       // (unreachable when U <= 3)
                        case 2: group(std::integral_constant<int, 2>{}); break;
                        case 1: group(std::integral_constant<int, 1>{}); break;
                        default: break;
                        }
                    }
                    if (bin_ok) {
                        float *t = tile + (4 * q) * ts + bin;
                        if (count_pow2) {      // x / 2^k == x * 2^-k bit for bit (both are the correctly rounded quotient)
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
            }
        } else if (i > 0) {
            // ---- store block cb0 + i - 1 out of tile[(i - 1) & 1]: kPipeCh * bins contiguous floats of out[r, c0 : c0 + kPipeCh, :, :]
            const float *tile = tiles + ((i - 1) & 1) * tile_floats;
            const int c0 = (cb0 + i - 1) * kPipeCh;
            const int cn = min(kPipeCh, C - c0);
            float *dst = out + (r * C + c0) * (int64_t)bins;
            if ((bins & 3) == 0) {
                // 16 bytes per lane: four consecutive bins of one channel; (channel, bin quad) advance incrementally
                const int qpc = bins >> 2;                    // quads per channel
                int c = 0, b4 = lane;
                while (b4 >= qpc) {
                    b4 -= qpc;
                    c++;
                }
                const int step_c = 64 / qpc, step_b = 64 - step_c * qpc;
                while (c < cn) {
                    const float *t = tile + c * ts + 4 * b4;
                    const float4 v = {t[0], t[1], t[2], t[3]};
                    store4_nt(dst + (c * bins + 4 * b4), v);
                    c += step_c;
                    b4 += step_b;
                    if (b4 >= qpc) {
                        b4 -= qpc;
                        c++;
                    }
                }
            } else {
                const float inv_bins = 1.0f / (float)bins;
                for (int idx = lane; idx < cn * bins; idx += 64) {
                    const int c = (int)(((float)idx + 0.5f) * inv_bins);       // idx / bins, exact for these sizes
                    dst[idx] = tile[c * ts + (idx - c * bins)];
                }
            }
        }
        // tile[i & 1] is complete and every LDS read of tile[(i - 1) & 1] has returned: LDS operations only -- a __syncthreads()
        // would also make the store wave wait for its global stores to RETIRE (the fence's vmcnt(0)), and the gather waves with it
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

bool roi_align_pipe_applicable(int C, int PH, int PW)
{
    static const int off = [] { const char *e = getenv("LOCOV_ROIALIGN_PIPE"); return e && atoi(e) == 0; }();
    if (off) return false;
    const int bins = PH * PW, ts = bins | 1;
    const size_t lds = 2 * (((size_t)kPipeCh * ts + 3) & ~(size_t)3) * sizeof(float) + 2 * kPipeAxis * sizeof(AxisSampleN);
    return C % 4 == 0 && C >= 4 * kPipeCh && lds <= 64 * 1024;
}

int launch_roi_align_pipe(const float *feat_nhwc, int N, int H, int W, int C, const float *rois, int64_t R, int PH, int PW, float scale,
                          int sampling_ratio, int aligned, float *out, hipStream_t s)
{
    const int bins = PH * PW, ts = bins | 1;
    const size_t lds = 2 * (((size_t)kPipeCh * ts + 3) & ~(size_t)3) * sizeof(float) + 2 * kPipeAxis * sizeof(AxisSampleN);
    const int nblk = (C + kPipeCh - 1) / kPipeCh;
    // enough workgroups for several rounds over the chip even at a small ROI count: split a ROI's channel blocks over grid.y
    int64_t splits = ceil_div((int64_t)4096, R);
    splits = splits < 1 ? 1 : splits > nblk ? nblk : splits;
    const int per = (int)ceil_div((int64_t)nblk, splits);
    dim3 grid((unsigned)R, (unsigned)ceil_div((int64_t)nblk, (int64_t)per));
    hipLaunchKernelGGL(roi_align_pipe_kernel, grid, dim3(kPipeThreads), lds, s, feat_nhwc, N, H, W, C, rois, PH, PW, scale,
                       sampling_ratio, aligned, out, per);
    return check_launch("locov_roi_align_from_nhwc_fwd (pipelined)");
}

}  // namespace locov
