// 3x3 / pad 1 / stride 1 convolution of 7x7 position-major tiles in the minimal-filtering
// (Toom-Cook / Winograd) domain -- the Res5 conv2 of [D2-upstream] BottleneckBlock as built by
// roi_emb_heads.py:217-241 and applied at :245,:323.
//
// A row of 7 outputs is split into an F(4,3) and an F(3,3) segment: 11 products per axis, 121 per
// (input channel, output channel, ROI) instead of the 361 real taps of the direct form (441 with
// the padding taps), i.e. 3x fewer matrix-core FLOPs for the layer that is 42 % of the stage:
//
//   V[f][r][c] = (BT (x) BT) x[.][r][c]          input transform   (HBM-bound, this file)
//   M[f][r][n] = sum_c V[f][r][c] * U[f][n][c]   121 GEMMs [R,Cin]x[N,Cin]^T in ONE batched launch (gemm_nt.hip)
//   y[.][r][n] = epi((AT (x) AT) M[.][r][n])     output transform + FrozenBN scale/shift + ReLU
//
// with f = fy*11 + fx and U = (G (x) G) w computed once, in fp64, when the weights are packed.
// BT and AT are small integers (exact in fp32); tables and their derivation: winograd_tables.h /
// tools/gen_winograd_tables.py.  fp32 error of the whole stage vs the direct form stays ~5e-6 of
// the activation range (tests/test_gpu_winograd.py; logits gate 1e-4).
//
// The transforms keep two channels per lane (8-byte accesses, 512 contiguous bytes per wave and
// row) so that a 7x7 patch (49 x 2 registers) or the 7x7 accumulators fit without spilling.
#include "winograd_transform.h"
#include "winograd_filter.h"

#include <cstdlib>

#ifndef LOCOV_WINO_THREADS
#define LOCOV_WINO_THREADS 256
#endif
// minimum waves per SIMD asked of the register allocator for the output / input transforms (second __launch_bounds__ argument)
#ifndef LOCOV_WINO_OUT_MINW
#define LOCOV_WINO_OUT_MINW 1
#endif
#ifndef LOCOV_WINO_IN_MINW
#define LOCOV_WINO_IN_MINW 1
#endif

namespace locov {

using wino::AT;
using wino::BT;
using wino::G;
using wino::NF;

// w [N, Cin, 3, 3] -> U [NF*NF, N, Cin]
__global__ __launch_bounds__(256) void wino_pack_weight_kernel(const float *__restrict__ w, int64_t NC,
                                                               float *__restrict__ U)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    double g[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) g[a][b] = (double)w[i * 9 + a * 3 + b];
#pragma unroll
    for (int fy = 0; fy < NF; fy++) {
        double t[3];
#pragma unroll
        for (int b = 0; b < 3; b++) t[b] = wino::filter_dot3(fy, g[0][b], g[1][b], g[2][b]);
#pragma unroll
        for (int fx = 0; fx < NF; fx++)
            U[(int64_t)(fy * NF + fx) * NC + i] = (float)wino::filter_dot3(fx, t[0], t[1], t[2]);
    }
}

// x rows [(y*7+x)*ld_pos + r*ld_roi][C]  ->  V [NF*NF][Rc][C]   (position-major input: ld_pos = R, ld_roi = 1;
// ROI-major input, row = roi*49 + position: ld_pos = 1, ld_roi = 49)
// SPLIT: V is written in the split layout scaled by v_scale (the A operand of the split batched GEMM); a value outside fp16's
// range raises *overflow (the GEMM no longer sees the fp32 values: the range guard moves here).
// amax_out (fp32 V only): a 16-byte operand-scale slot whose word 2 receives max |V| (gemm_nt.h, amax_fold) -- the transform of
// a gradient has no a-priori range, the GEMM that reads V derives its operand scale from that word.
template <bool GRAD, bool SPLIT = false>
__global__ __launch_bounds__(256, LOCOV_WINO_IN_MINW) void wino_input_kernel(const float *__restrict__ x, int64_t ld_pos, int64_t ld_roi, int64_t Rc, int C,
                                                         float *__restrict__ V, float v_scale = 1.f, unsigned *overflow = nullptr,
                                                         float *amax_out = nullptr)
{
    const int c2 = C >> 1;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (no early return: the whole wave meets in amax_fold.  SPLIT: lanes trade words in pairs (t, t^1); C % 4 == 0 makes the count
    //  even, so a pair is live or idle as a whole)
    const bool live = t < Rc * c2;
    float amax = 0.f;
    if (live) {
    const int64_t r = t / c2;
    const int c = (int)(t - r * c2) * 2;
    const float *src = x + r * ld_roi * C + c;
    f32x2 d[7][7];
#pragma unroll
    for (int y = 0; y < 7; y++)
#pragma unroll
        for (int xx = 0; xx < 7; xx++)
            d[y][xx] = *reinterpret_cast<const f32x2 *>(src + (int64_t)(y * 7 + xx) * ld_pos * C);
    float *dst = V + r * C + c;
    const int64_t fstride = Rc * C;
    wino_in_all<GRAD>([&](int y, int xx) { return d[y][xx]; },
                      [&](int fy, int fx, f32x2 a) {
                          amax = fmaxf(fmaxf(amax, fabsf(a[0])), fabsf(a[1]));
                          if constexpr (SPLIT)
                              store_split_pair(V + r * C + (int64_t)(fy * NF + fx) * fstride, c, a, v_scale);
                          else
                              wino_store(a, reinterpret_cast<f32x2 *>(dst + (int64_t)(fy * NF + fx) * fstride));
                      });
    }
    if (SPLIT && overflow != nullptr && amax * v_scale >= 65504.f) atomicOr(overflow, 1u);
    if (!SPLIT && amax_out != nullptr) amax_fold(amax_out, amax);
}

// M [NF*NF][Rc][N]  ->  y rows [(y*7+x)*ld_pos + r*ld_roi] (ldy elements apart) = relu?(acc * scale[n] + shift[n]);
// position-major output: ld_pos = R, ld_roi = 1; ROI-major output (row = roi*49 + position): ld_pos = 1, ld_roi = 49
// SPLIT: y is written in the split layout scaled by y_scale (it is then the pre-split A operand of the 1x1 convolution that
// follows); a finished value outside fp16's range raises *overflow.
// WIDE = false (the launcher's choice whenever the 11 planes of a row of M and the 49 rows of a ROI's output stay within 32-bit byte
// offsets): every access goes through a buffer descriptor with the plane / position as a SCALAR offset and the lane's (ROI, channel
// pair) as one 32-bit vector offset.  As flat accesses the kernel carried 121 + 49 64-bit addresses in VGPRs: 256 + 20 registers, one
// wave per SIMD on a kernel that lives on memory-level parallelism; now 254 and two (8 000 ROIs, 512 channels: 540 -> 485 us).  The
// compiler still issues all 121 loads before the first add -- scheduling fences between the rows of planes and a lower register budget
// (LOCOV_WINO_OUT_MINW) only turn the loaded values into scratch traffic -- which at two waves per SIMD is what keeps HBM busy.
// MASKED (the data gradient of a 3x3 convolution: the ReLU backward of the saved activation in the epilogue): the 49 mask values
// of the lane's (ROI, channel pair) are requested TOGETHER once the accumulators are final -- the transform's operands are dead by
// then, so the registers are there.  Left inside the store loop every position's load sat between two stores the compiler may not
// move it across (49 dependent round trips per lane: 142 us per launch against the forward's 57 us at 800 ROIs).
template <bool SPLIT = false, bool WIDE = false, bool MASKED = false>
__global__ __launch_bounds__(256, LOCOV_WINO_OUT_MINW) void wino_output_kernel(const float *__restrict__ Mv, int64_t ld_pos, int64_t ld_roi, int64_t Rc, int N,
                                                          const float *__restrict__ scale,
                                                          const float *__restrict__ shift, int relu,
                                                          float *__restrict__ y, int64_t ldy, const float *__restrict__ mask,
                                                          float y_scale = 1.f, unsigned *overflow = nullptr, float *amax_out = nullptr)
{
    // mask (same rows and pitch as y, or null): the value is kept where mask > 0, zeroed elsewhere -- the ReLU backward of the
    // saved activation when this convolution is a data gradient (flipped filter)
    const int n2 = N >> 1;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < Rc * n2;       // (no early return: the whole wave meets in amax_fold)
    float amax = 0.f;
    if (live) {
    const int64_t r = t / n2;
    const int n = (int)(t - r * n2) * 2;
    const float *src = Mv + r * N + n;
    const int64_t fstride = Rc * N;
    const unsigned lane_off = (unsigned)((r * N + n) * 4), fbytes = (unsigned)(fstride * 4);
    f32x2 acc[7][7];
#pragma unroll
    for (int yy = 0; yy < 7; yy++)
#pragma unroll
        for (int xx = 0; xx < 7; xx++) acc[yy][xx] = f32x2{0.f, 0.f};
#pragma unroll
    for (int fy = 0; fy < NF; fy++) {
        f32x2 m[NF];
        if constexpr (!WIDE) {
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Mv + (int64_t)fy * NF * fstride), 0, 0xffffffff, 0x00020000);
#pragma unroll
            for (int fx = 0; fx < NF; fx++)
                m[fx] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rm, lane_off, (unsigned)fx * fbytes, LOCOV_WINO_NT_LOAD ? 2 : 0));
        } else {
#pragma unroll
            for (int fx = 0; fx < NF; fx++)
                m[fx] = wino_load(reinterpret_cast<const f32x2 *>(src + (int64_t)(fy * NF + fx) * fstride));
        }
        f32x2 tx[7];
#pragma unroll
        for (int xx = 0; xx < 7; xx++) {
            f32x2 a = {0.f, 0.f};
#pragma unroll
            for (int fx = 0; fx < NF; fx++)
                if (AT[xx][fx] != 0.f) a += AT[xx][fx] * m[fx];
            tx[xx] = a;
        }
#pragma unroll
        for (int yy = 0; yy < 7; yy++)
            if (AT[yy][fy] != 0.f) {
#pragma unroll
                for (int xx = 0; xx < 7; xx++) acc[yy][xx] += AT[yy][fy] * tx[xx];
            }
    }
    f32x2 sc = {1.f, 1.f}, sh = {0.f, 0.f};
    if (scale) sc = *reinterpret_cast<const f32x2 *>(scale + n);
    if (shift) sh = *reinterpret_cast<const f32x2 *>(shift + n);
    float *dst = y + r * ld_roi * ldy + n;
    const float *msk = mask ? mask + r * ld_roi * ldy + n : nullptr;
    // narrow form: the ROI's row block as the vector offset, the position as a scalar one
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(mask ? mask : y), 0, 0xffffffff, 0x00020000);
    const unsigned row_off = (unsigned)(r * ld_roi * ldy * 4), pos_bytes = (unsigned)(ld_pos * ldy * 4);
    const unsigned y_off = row_off + (SPLIT ? (unsigned)split_pair_offset(n) : (unsigned)n * 4u);
    f32x2 mk[MASKED ? 7 : 1][MASKED ? 7 : 1];
    if constexpr (MASKED) {
#pragma unroll
        for (int yy = 0; yy < 7; yy++)
#pragma unroll
            for (int xx = 0; xx < 7; xx++)
                mk[yy][xx] = WIDE ? *reinterpret_cast<const f32x2 *>(msk + (int64_t)(yy * 7 + xx) * ld_pos * ldy)
                                  : __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rk, row_off + (unsigned)n * 4u,
                                                                                                   (unsigned)(yy * 7 + xx) * pos_bytes, 0));
    }
#pragma unroll
    for (int yy = 0; yy < 7; yy++) {
#pragma unroll
        for (int xx = 0; xx < 7; xx++) {
            f32x2 v = acc[yy][xx] * sc + sh;
            if (relu) {
                v[0] = fmaxf(v[0], 0.f);
                v[1] = fmaxf(v[1], 0.f);
            }
            if constexpr (MASKED) {                                // (a split-layout output takes no mask: locov_winograd_conv3x3_f32_split_ex)
                v[0] = mk[yy][xx][0] > 0.f ? v[0] : 0.f;
                v[1] = mk[yy][xx][1] > 0.f ? v[1] : 0.f;
            }
            amax = fmaxf(fmaxf(amax, fabsf(v[0])), fabsf(v[1]));
            if constexpr (SPLIT) {
                if constexpr (WIDE)
                    store_split_pair(y + r * ld_roi * ldy + (int64_t)(yy * 7 + xx) * ld_pos * ldy, n, v, y_scale);
                else
                    __builtin_amdgcn_raw_buffer_store_b64(split_pair_words(n, v, y_scale), ry, y_off, (unsigned)(yy * 7 + xx) * pos_bytes,
                                                          LOCOV_WINO_NT_STORE ? 2 : 0);
            } else {
                if constexpr (WIDE)
                    *reinterpret_cast<f32x2 *>(dst + (int64_t)(yy * 7 + xx) * ld_pos * ldy) = v;
                else
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(wino_u32x2, v), ry, y_off, (unsigned)(yy * 7 + xx) * pos_bytes, 0);
            }
        }
    }
    }
    // (SPLIT: what was just written no longer holds the fp32 value.  fp32 output with `overflow`: the caller asks for the range check
    //  of the split GEMM that will read y at operand scale y_scale -- the guard word is then final one launch earlier)
    if (overflow != nullptr && amax * y_scale >= 65504.f) atomicOr(overflow, 1u);
    if (!SPLIT && amax_out != nullptr) amax_fold(amax_out, amax);
}

// dU [NF*NF, N, Cin] -> dw [N, Cin, 3, 3] = row_scale[n] * (G (x) G)^T dU : the adjoint of wino_pack_weight_kernel
__global__ __launch_bounds__(256) void wino_unpack_wgrad_kernel(const float *__restrict__ dU, int64_t NC, int Cin,
                                                                const float *__restrict__ row_scale, float *__restrict__ dw)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    double g[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int fy = 0; fy < NF; fy++) {
        double t[3] = {0, 0, 0};
#pragma unroll
        for (int fx = 0; fx < NF; fx++) {
            const double u = (double)dU[(int64_t)(fy * NF + fx) * NC + i];
#pragma unroll
            for (int b = 0; b < 3; b++) t[b] += G[fx][b] * u;
        }
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) g[a][b] += G[fy][a] * t[b];
    }
    const double sc = row_scale ? (double)row_scale[i / Cin] : 1.0;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) dw[i * 9 + a * 3 + b] = (float)(sc * g[a][b]);
}

// ROIs per pass: the two transform-domain buffers of a pass (121 * chunk * (Cin + N) floats) are
// sized to stay inside the 256 MB Infinity Cache between the three kernels that produce / consume them.
static int64_t chunk_rois(int64_t R, int Cin, int N)
{
    static const int64_t forced = [] {
        const char *e = getenv("LOCOV_WINO_CHUNK");
        return e ? (int64_t)atoll(e) : (int64_t)-1;
    }();
    int64_t c = forced >= 0 ? forced : 0;
    if (c <= 0 || c > R) c = R;
    (void)Cin;
    (void)N;
    return c;
}

}  // namespace locov

using namespace locov;

extern "C" {

int64_t locov_winograd_workspace_bytes(int64_t R, int Cin, int N)
{
    if (R <= 0 || Cin <= 0 || N <= 0) return 0;
    const int64_t c = chunk_rois(R, Cin, N);
    return (int64_t)NF * NF * c * ((int64_t)Cin + N) * (int64_t)sizeof(float) + 16;     // + the device-chosen operand scale (split_ex form)
}

int locov_winograd_pack_weight(const float *w, int N, int Cin, float *U, locov_stream_t stream)
{
    LOCOV_REQUIRE(N > 0 && Cin > 0, "locov_winograd_pack_weight: bad shape");
    LOCOV_REQUIRE(w && U, "locov_winograd_pack_weight: null pointer");
    const int64_t NC = (int64_t)N * Cin;
    hipLaunchKernelGGL(wino_pack_weight_kernel, dim3((unsigned)ceil_div(NC, 256)), dim3(256), 0, as_stream(stream), w, NC, U);
    return check_launch("locov_winograd_pack_weight");
}

// the 1x1 convolution (+ FrozenBN + ReLU) in front of the 3x3 one, when both are asked for in one call
// (locov_conv1x1_winograd_conv3x3_f32_split): its input in the split layout, ROI-major rows; y1 = pixel scratch for the unfused form
struct PreConv {
    const float *x;
    int64_t ldx;
    int K;
    float x_scale;
    const void *W;
    float w_scale;
    const float *scale, *shift;
    float *y1;
};

// ... or the pooler in front of it (locov_roi_align_winograd_conv3x3_f32_split: block 0, whose 1x1 convolution ran on the map):
// even bins of a `pooled` x `pooled` ROIAlign of a channels-last fp32 map + per-channel affine + ReLU, ROI-major rows
struct PrePool {
    const float *feat;
    int N, H, W;
    int64_t feat_ld;
    const float *rois;
    int pooled;
    float spatial_scale;
    int sampling_ratio, aligned;
    const float *scale, *shift;
    float *y1;
};

// u_scale > 0: U is the split-operand packing of u_scale * U (locov_split_f16x2_pack) and the 121 GEMMs run on the
// f16 matrix pipe with V scaled by v_scale (gemm_split.hip); u_scale == 0: fp32 U, fp32 MFMA.
static int winograd_conv3x3(const float *x, int64_t R, int Cin, const float *U, float u_scale, float v_scale,
                            const float *scale, const float *shift, float *y, int64_t ldy, int N, unsigned flags,
                            void *workspace, int64_t workspace_bytes, locov_stream_t stream, const float *mask = nullptr,
                            unsigned *overflow = nullptr, bool v_scale_auto = false, float y_split_scale = 0.f,
                            float *amax_out = nullptr, const PreConv *pre = nullptr, const PrePool *pool = nullptr);

int locov_winograd_conv3x3_f32_ex(const float *x, int64_t R, int Cin, const float *U, const float *scale, const float *shift,
                                  const float *mask, float *y, int64_t ldy, int N, unsigned flags, void *workspace,
                                  int64_t workspace_bytes, locov_stream_t stream)
{
    return winograd_conv3x3(x, R, Cin, U, 0.f, 0.f, scale, shift, y, ldy, N, flags, workspace, workspace_bytes, stream, mask);
}

int locov_winograd_conv3x3_f32_split_ex(const float *x, int64_t R, int Cin, const void *U_split, float u_scale, float v_scale,
                                        int v_scale_auto, const float *scale, const float *shift, const float *mask, float *y,
                                        int64_t ldy, int N, unsigned flags, float y_split_scale, void *workspace,
                                        int64_t workspace_bytes, unsigned *overflow, float *amax_out, locov_stream_t stream)
{
    LOCOV_REQUIRE(u_scale > 0.f && (v_scale_auto || v_scale > 0.f), "locov_winograd_conv3x3_f32_split_ex: operand scales must be positive");
    LOCOV_REQUIRE(!amax_out || !(y_split_scale > 0.f), "locov_winograd_conv3x3_f32_split_ex: amax_out is for an fp32 output");
    return winograd_conv3x3(x, R, Cin, static_cast<const float *>(U_split), u_scale, v_scale_auto ? 1.f : v_scale, scale, shift, y, ldy, N,
                            flags, workspace, workspace_bytes, stream, mask, overflow, v_scale_auto != 0, y_split_scale, amax_out);
}

int locov_winograd_conv3x3_f32(const float *x, int64_t R, int Cin, const float *U, const float *scale,
                               const float *shift, float *y, int64_t ldy, int N, unsigned flags, void *workspace,
                               int64_t workspace_bytes, locov_stream_t stream)
{
    return winograd_conv3x3(x, R, Cin, U, 0.f, 0.f, scale, shift, y, ldy, N, flags, workspace, workspace_bytes, stream);
}

int locov_winograd_conv3x3_f32_split(const float *x, int64_t R, int Cin, const void *U_split, float u_scale, float v_scale,
                                     const float *scale, const float *shift, float *y, int64_t ldy, int N,
                                     unsigned flags, void *workspace, int64_t workspace_bytes, unsigned *overflow,
                                     locov_stream_t stream)
{
    LOCOV_REQUIRE(u_scale > 0.f && v_scale > 0.f, "locov_winograd_conv3x3_f32_split: operand scales must be positive");
    return winograd_conv3x3(x, R, Cin, static_cast<const float *>(U_split), u_scale, v_scale, scale, shift, y, ldy, N, flags,
                            workspace, workspace_bytes, stream, nullptr, overflow);
}

int locov_roi_align_winograd_conv3x3_f32_split(const float *feat_nhwc, int Nimg, int H, int W, int C, int64_t feat_ld, const float *rois,
                                               int64_t R, int pooled, float spatial_scale, int sampling_ratio, int aligned,
                                               const float *scale1, const float *shift1, const void *U_split, float u_scale, float v_scale,
                                               const float *scale2, const float *shift2, float *y, int64_t ldy, int N, unsigned flags,
                                               float y_split_scale, void *workspace, int64_t workspace_bytes, unsigned *overflow,
                                               locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && Nimg > 0 && H > 0 && W > 0 && C > 0 && N > 0, "locov_roi_align_winograd_conv3x3_f32_split: bad shape");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(feat_nhwc && rois && U_split && y && workspace, "locov_roi_align_winograd_conv3x3_f32_split: null pointer");
    LOCOV_REQUIRE(pooled == 14 || pooled == 13, "locov_roi_align_winograd_conv3x3_f32_split: the even bins of the pooler must form a 7 x 7 tile");
    LOCOV_REQUIRE(u_scale > 0.f && v_scale > 0.f && spatial_scale > 0.f, "locov_roi_align_winograd_conv3x3_f32_split: scales must be positive");
    LOCOV_REQUIRE(feat_ld >= C && feat_ld % 4 == 0 && C % 4 == 0 && (uintptr_t)feat_nhwc % 16 == 0 && (int64_t)H * W * feat_ld * 4 < 0xffffffffLL,
                  "locov_roi_align_winograd_conv3x3_f32_split: feat_ld >= C, both %% 4, 16-byte pointer, one image below 4 GiB");
    LOCOV_REQUIRE((flags & LOCOV_WINO_IN_ROI_MAJOR) != 0, "locov_roi_align_winograd_conv3x3_f32_split: rows are ROI-major (LOCOV_WINO_IN_ROI_MAJOR)");
    LOCOV_REQUIRE(chunk_rois(R, C, N) == R, "locov_roi_align_winograd_conv3x3_f32_split: one pass only (LOCOV_WINO_CHUNK is set)");
    const int64_t wb = locov_winograd_workspace_bytes(R, C, N);
    LOCOV_REQUIRE(workspace_bytes >= locov_roi_align_winograd_workspace_bytes(R, C, N),
                  "locov_roi_align_winograd_conv3x3_f32_split: workspace too small (%lld bytes)", (long long)workspace_bytes);
    PrePool pool{feat_nhwc, Nimg, H, W, feat_ld, rois, pooled, spatial_scale, sampling_ratio, aligned, scale1, shift1,
                 reinterpret_cast<float *>(static_cast<char *>(workspace) + ((wb + 15) & ~(int64_t)15))};
    return winograd_conv3x3(nullptr, R, C, static_cast<const float *>(U_split), u_scale, v_scale, scale2, shift2, y, ldy, N, flags, workspace, wb,
                            stream, nullptr, overflow, false, y_split_scale, nullptr, nullptr, &pool);
}

int64_t locov_conv1x1_winograd_workspace_bytes(int64_t R, int C, int N)
{
    if (R <= 0 || C <= 0 || N <= 0) return 0;
    return locov_winograd_workspace_bytes(R, C, N) + R * 49 * (int64_t)C * (int64_t)sizeof(float);
}

// What the two fused calls need for THESE shapes: the [49 R, C] pixel scratch behind the transform-domain workspace exists only
// for the two-launch fallback -- where the producer writes the Winograd input transform itself nothing ever touches it
// (0.8 GB at 8 000 ROIs that a cached workspace would otherwise hold for good).
static bool conv1x1_wino_fused(int64_t R, int K, int C, int64_t ldx)
{
    return gemm_split_big_wino_applicable(ldx, R * 49, C, K, Epilogue{nullptr, nullptr, nullptr, LOCOV_EPI_RELU | LOCOV_GEMM_A_SPLIT});
}

int64_t locov_conv1x1_winograd_workspace_bytes_for(int64_t R, int K, int C, int N, int64_t ldx)
{
    if (R <= 0 || C <= 0 || N <= 0 || K <= 0) return 0;
    const int64_t wb = locov_winograd_workspace_bytes(R, C, N);
    return chunk_rois(R, C, N) == R && conv1x1_wino_fused(R, K, C, ldx) ? wb : locov_conv1x1_winograd_workspace_bytes(R, C, N);
}

int64_t locov_roi_align_winograd_workspace_bytes(int64_t R, int C, int N)
{
    if (R <= 0 || C <= 0 || N <= 0) return 0;
    const int64_t wb = locov_winograd_workspace_bytes(R, C, N);
    return chunk_rois(R, C, N) == R && roi_align_nhwc_wino_applicable(C, R) ? wb : locov_conv1x1_winograd_workspace_bytes(R, C, N);
}

int locov_conv1x1_winograd_conv3x3_f32_split(const float *x_split, int64_t ldx, int K, float x_scale, const void *W1_split, float w1_scale,
                                             const float *scale1, const float *shift1, int64_t R, int C, const void *U_split,
                                             float u_scale, float v_scale, const float *scale2, const float *shift2, float *y,
                                             int64_t ldy, int N, unsigned flags, float y_split_scale, void *workspace,
                                             int64_t workspace_bytes, unsigned *overflow, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && K > 0 && C > 0 && N > 0, "locov_conv1x1_winograd_conv3x3_f32_split: bad shape");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x_split && W1_split && U_split && y && workspace, "locov_conv1x1_winograd_conv3x3_f32_split: null pointer");
    LOCOV_REQUIRE(x_scale > 0.f && w1_scale > 0.f && u_scale > 0.f && v_scale > 0.f,
                  "locov_conv1x1_winograd_conv3x3_f32_split: operand scales must be positive");
    LOCOV_REQUIRE(K % 32 == 0 && ldx >= K && ldx % 4 == 0 && (uintptr_t)x_split % 16 == 0 && (uintptr_t)W1_split % 16 == 0,
                  "locov_conv1x1_winograd_conv3x3_f32_split: K %% 32, ldx >= K, ldx %% 4 and 16-byte pointers required");
    LOCOV_REQUIRE((flags & LOCOV_WINO_IN_ROI_MAJOR) != 0, "locov_conv1x1_winograd_conv3x3_f32_split: rows are ROI-major (LOCOV_WINO_IN_ROI_MAJOR)");
    LOCOV_REQUIRE(chunk_rois(R, C, N) == R, "locov_conv1x1_winograd_conv3x3_f32_split: one pass only (LOCOV_WINO_CHUNK is set)");
    const int64_t wb = locov_winograd_workspace_bytes(R, C, N);
    LOCOV_REQUIRE(workspace_bytes >= locov_conv1x1_winograd_workspace_bytes_for(R, K, C, N, ldx),
                  "locov_conv1x1_winograd_conv3x3_f32_split: workspace too small (%lld bytes)", (long long)workspace_bytes);
    PreConv pre{x_split, ldx, K, x_scale, W1_split, w1_scale, scale1, shift1,
                reinterpret_cast<float *>(static_cast<char *>(workspace) + ((wb + 15) & ~(int64_t)15))};
    return winograd_conv3x3(nullptr, R, C, static_cast<const float *>(U_split), u_scale, v_scale, scale2, shift2, y, ldy, N, flags, workspace, wb,
                            stream, nullptr, overflow, false, y_split_scale, nullptr, &pre);
}

static int winograd_conv3x3(const float *x, int64_t R, int Cin, const float *U, float u_scale, float v_scale,
                            const float *scale, const float *shift, float *y, int64_t ldy, int N, unsigned flags,
                            void *workspace, int64_t workspace_bytes, locov_stream_t stream, const float *mask, unsigned *overflow,
                            bool v_scale_auto, float y_split_scale, float *amax_out, const PreConv *pre, const PrePool *pool)
{
    LOCOV_REQUIRE(ldy >= N && ldy % 2 == 0, "locov_winograd_conv3x3_f32: ldy must be >= N and even");
    LOCOV_REQUIRE(R >= 0 && Cin > 0 && N > 0, "locov_winograd_conv3x3_f32: bad shape");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE((x || pre || pool) && U && y && workspace, "locov_winograd_conv3x3_f32: null pointer");
    LOCOV_REQUIRE(Cin % 32 == 0 && N % 4 == 0, "locov_winograd_conv3x3_f32: Cin %% 32 and N %% 4 must be 0 (got %d, %d)",
                  Cin, N);
    LOCOV_REQUIRE(((uintptr_t)x | (uintptr_t)U | (uintptr_t)y | (uintptr_t)workspace) % 16 == 0,
                  "locov_winograd_conv3x3_f32: misaligned pointer");
    LOCOV_REQUIRE(!(flags & ~(unsigned)(LOCOV_EPI_RELU | LOCOV_WINO_OUT_ROI_MAJOR | LOCOV_WINO_IN_ROI_MAJOR)),
                  "locov_winograd_conv3x3_f32: unsupported flags 0x%x",
                  flags);
    LOCOV_REQUIRE(!(y_split_scale > 0.f) || (N % 32 == 0 && ldy == N && !mask),
                  "locov_winograd_conv3x3_f32_split: a split-layout output needs N %% 32 == 0, ldy == N and no mask");
    LOCOV_REQUIRE(workspace_bytes >= locov_winograd_workspace_bytes(R, Cin, N),
                  "locov_winograd_conv3x3_f32: workspace too small (%lld bytes)", (long long)workspace_bytes);
    const int64_t chunk = chunk_rois(R, Cin, N);
    float *V = static_cast<float *>(workspace);
    float *Mv = V + (int64_t)NF * NF * chunk * Cin;
    hipStream_t s = as_stream(stream);
    for (int64_t r0 = 0; r0 < R; r0 += chunk) {
        const int64_t rc = R - r0 < chunk ? R - r0 : chunk;
        const int64_t tin = rc * (Cin / 2), tout = rc * (N / 2);
        const bool in_roi_major = (flags & LOCOV_WINO_IN_ROI_MAJOR) != 0;
        // split GEMM with a scale known up front: V leaves the transform already in the split layout (staged by DMA in the GEMM)
        const bool v_split = u_scale > 0.f && !v_scale_auto;
        bool v_done = false;
        if (pre) {
            // (pre: split arithmetic, fixed v_scale, ROI-major rows, one pass -- checked by the caller)
            Epilogue e1{pre->scale, pre->shift, nullptr, LOCOV_EPI_RELU | LOCOV_GEMM_A_SPLIT};
            int rc1;
            if (gemm_split_big_wino_applicable(pre->ldx, rc * 49, Cin, pre->K, e1)) {
                rc1 = launch_gemm_split_big_wino(pre->x, pre->ldx, pre->W, rc * 49, Cin, pre->K, e1, pre->x_scale, pre->w_scale, V, v_scale, s,
                                                 "locov_conv1x1_winograd_conv3x3_f32_split (1x1 + input transform)", overflow);
                v_done = true;
            } else {
                rc1 = launch_gemm_split(pre->x, pre->ldx, pre->W, pre->y1, (int64_t)Cin, rc * 49, Cin, pre->K, e1, pre->x_scale, pre->w_scale, s,
                                        "locov_conv1x1_winograd_conv3x3_f32_split (1x1)", Batch{1, 0, 0, 0}, overflow);
                x = pre->y1;
            }
            if (rc1) return rc1;
        }
        if (pool) {
            int rc1;
            if (roi_align_nhwc_wino_applicable(Cin, rc)) {
                rc1 = launch_roi_align_nhwc_wino(pool->feat, pool->N, pool->H, pool->W, Cin, pool->feat_ld, pool->rois, rc, pool->pooled,
                                                 pool->spatial_scale, pool->sampling_ratio, pool->aligned, pool->scale, pool->shift, 1, V, v_scale,
                                                 overflow, s);
                v_done = true;
            } else {
                rc1 = locov_roi_align_nhwc_affine_fwd(pool->feat, LOCOV_F32, pool->N, pool->H, pool->W, Cin, pool->feat_ld, pool->rois, rc,
                                                      pool->pooled, pool->pooled, pool->spatial_scale, pool->sampling_ratio, pool->aligned, 2, 0,
                                                      pool->scale, pool->shift, 1, pool->y1, (int64_t)Cin, LOCOV_F32, stream);
                x = pool->y1;
            }
            if (rc1) return rc1;
        }
        if (v_done)
            ;
        else if (v_split)
            hipLaunchKernelGGL((wino_input_kernel<false, true>), dim3((unsigned)ceil_div(tin, 256)), dim3(256), 0, s,
                               x + r0 * (in_roi_major ? 49 : 1) * Cin, in_roi_major ? (int64_t)1 : R,
                               in_roi_major ? (int64_t)49 : (int64_t)1, rc, Cin, V, v_scale, overflow);
        else {
            // (split GEMM on a gradient: the transform folds max |V| into the 16-byte slot at the end of the workspace, zeroed
            // here, and the GEMM derives V's operand scale from it -- no separate pass over the 121 x rc x Cin values)
            float *scw = u_scale > 0.f && v_scale_auto ? reinterpret_cast<float *>(static_cast<char *>(workspace) + workspace_bytes - 16) : nullptr;
            if (scw && hipMemsetAsync(scw, 0, 16, s) != hipSuccess) return set_error(LOCOV_ERR_LAUNCH, "locov_winograd_conv3x3_f32_split: memset failed");
            hipLaunchKernelGGL((wino_input_kernel<false, false>), dim3((unsigned)ceil_div(tin, 256)), dim3(256), 0, s,
                               x + r0 * (in_roi_major ? 49 : 1) * Cin, in_roi_major ? (int64_t)1 : R,
                               in_roi_major ? (int64_t)49 : (int64_t)1, rc, Cin, V, 1.f, static_cast<unsigned *>(nullptr), scw);
        }
        int rcode = check_launch("locov_winograd_conv3x3_f32 (input transform)");
        if (rcode) return rcode;
        Epilogue epi{nullptr, nullptr, nullptr, v_split ? LOCOV_GEMM_A_SPLIT : 0u};
        if (u_scale > 0.f) {
            const float *sc = nullptr;
            if (v_scale_auto)         // the input is a gradient: the scale of its transform comes from the slot the transform filled
                sc = reinterpret_cast<float *>(static_cast<char *>(workspace) + workspace_bytes - 16);
            rcode = launch_gemm_split(V, (int64_t)Cin, U, Mv, (int64_t)N, rc, N, Cin, epi, v_scale, u_scale, s,
                                      "locov_winograd_conv3x3_f32_split (batched GEMM)",
                                      Batch{NF * NF, rc * Cin, (int64_t)N * Cin, rc * N}, overflow, sc);
        }
        else
            rcode = launch_gemm_nt<float, float>(V, (int64_t)Cin, U, (int64_t)Cin, Mv, (int64_t)N, rc, N, Cin, epi, s,
                                                 "locov_winograd_conv3x3_f32 (batched GEMM)", ConvGeom{0, 0, 0, 0, 0},
                                                 Batch{NF * NF, rc * Cin, (int64_t)N * Cin, rc * N});
        if (rcode) return rcode;
        const bool roi_major = (flags & LOCOV_WINO_OUT_ROI_MAJOR) != 0;
        // 32-bit byte offsets suffice for a row of 11 planes of M and for the 49 rows of every ROI of y (else the flat-address instance)
        // (position-major y: a position's rows are R -- the whole call's ROI count, not this chunk's rc -- apart)
        const bool wide = (uint64_t)rc * N * 4u * NF > 0xffffffffull ||
                          (uint64_t)(roi_major ? rc : R) * 49u * (uint64_t)ldy * 4u > 0xffffffffull;
        const int64_t lp = roi_major ? (int64_t)1 : R, lr = roi_major ? (int64_t)49 : (int64_t)1;
        float *yo = y + r0 * (roi_major ? 49 : 1) * ldy;
        const float *mo = mask ? mask + r0 * (roi_major ? 49 : 1) * ldy : nullptr;
        const int relu_i = (flags & LOCOV_EPI_RELU) ? 1 : 0;
        const dim3 og((unsigned)ceil_div(tout, 256));
#define LOCOV_WINO_OUT(SP, WD, ...) hipLaunchKernelGGL((wino_output_kernel<SP, WD>), og, dim3(256), 0, s, Mv, lp, lr, rc, N, scale, shift, relu_i, yo, ldy, __VA_ARGS__)
#define LOCOV_WINO_OUT_MASKED(WD, ...) hipLaunchKernelGGL((wino_output_kernel<false, WD, true>), og, dim3(256), 0, s, Mv, lp, lr, rc, N, scale, shift, relu_i, yo, ldy, __VA_ARGS__)
        if (y_split_scale > 0.f) {
            if (wide) LOCOV_WINO_OUT(true, true, static_cast<const float *>(nullptr), y_split_scale, overflow, static_cast<float *>(nullptr));
            else LOCOV_WINO_OUT(true, false, static_cast<const float *>(nullptr), y_split_scale, overflow, static_cast<float *>(nullptr));
        } else if (mo != nullptr) {
            if (wide) LOCOV_WINO_OUT_MASKED(true, mo, 1.f, static_cast<unsigned *>(nullptr), amax_out);
            else LOCOV_WINO_OUT_MASKED(false, mo, 1.f, static_cast<unsigned *>(nullptr), amax_out);
        } else {
            // y_split_scale < 0: fp32 output, range-checked at operand scale -y_split_scale
            unsigned *const chk = y_split_scale < 0.f ? overflow : nullptr;
            const float chk_scale = y_split_scale < 0.f ? -y_split_scale : 1.f;
            if (wide) LOCOV_WINO_OUT(false, true, mo, chk_scale, chk, amax_out);
            else LOCOV_WINO_OUT(false, false, mo, chk_scale, chk, amax_out);
        }
#undef LOCOV_WINO_OUT_MASKED
#undef LOCOV_WINO_OUT
        rcode = check_launch("locov_winograd_conv3x3_f32 (output transform)");
        if (rcode) return rcode;
    }
    return LOCOV_OK;
}

int64_t locov_winograd_wgrad_workspace_bytes(int64_t R, int Cin, int N)
{
    if (R <= 0 || Cin <= 0 || N <= 0) return 0;
    return (int64_t)NF * NF * (R * ((int64_t)Cin + N) + (int64_t)N * Cin) * (int64_t)sizeof(float) +
           gemm_tn_workspace_bytes(R, N, Cin, NF * NF) + 16;      // + {scale, 1/scale, amax bits} of the split form
}

static int winograd_wgrad(const float *x, const float *g, int64_t R, int Cin, int N, unsigned flags, const float *row_scale,
                          float *dw, void *workspace, int64_t workspace_bytes, locov_stream_t stream, bool split, unsigned *overflow,
                          const float *v_split = nullptr);

int locov_winograd_wgrad_f32(const float *x, const float *g, int64_t R, int Cin, int N, unsigned flags, const float *row_scale,
                             float *dw, void *workspace, int64_t workspace_bytes, locov_stream_t stream)
{
    return winograd_wgrad(x, g, R, Cin, N, flags, row_scale, dw, workspace, workspace_bytes, stream, false, nullptr);
}

int locov_winograd_wgrad_f32_split(const float *x, const float *g, int64_t R, int Cin, int N, unsigned flags, const float *row_scale,
                                   float *dw, unsigned *overflow, void *workspace, int64_t workspace_bytes, locov_stream_t stream)
{
    return winograd_wgrad(x, g, R, Cin, N, flags, row_scale, dw, workspace, workspace_bytes, stream, true, overflow);
}

int locov_winograd_wgrad_f32_split_v(const float *v_split, const float *g, int64_t R, int Cin, int N, unsigned flags, const float *row_scale,
                                     float *dw, unsigned *overflow, void *workspace, int64_t workspace_bytes, locov_stream_t stream)
{
    LOCOV_REQUIRE(v_split != nullptr || R == 0, "locov_winograd_wgrad_f32_split_v: null transformed input");
    LOCOV_REQUIRE(Cin % 8 == 0, "locov_winograd_wgrad_f32_split_v: Cin must be a multiple of 8 (got %d)", Cin);
    return winograd_wgrad(v_split, g, R, Cin, N, flags, row_scale, dw, workspace, workspace_bytes, stream, true, overflow, v_split);
}

// v_split: the forward's transformed input (wino_input_kernel<false, true>: [121][R][Cin] in the split layout at scale 0.25) kept by the
// caller -- x is then not transformed again and the TN GEMMs stage it as it is (gemm_tn_split.hip, B_SPLIT: the same bits)
static int winograd_wgrad(const float *x, const float *g, int64_t R, int Cin, int N, unsigned flags, const float *row_scale,
                          float *dw, void *workspace, int64_t workspace_bytes, locov_stream_t stream, bool split, unsigned *overflow,
                          const float *v_split)
{
    LOCOV_REQUIRE(R >= 0 && Cin > 0 && N > 0, "locov_winograd_wgrad_f32: bad shape");
    LOCOV_REQUIRE(dw, "locov_winograd_wgrad_f32: null output");
    hipStream_t s = as_stream(stream);
    if (R == 0) {
        hipError_t e = hipMemsetAsync(dw, 0, (size_t)N * Cin * 9 * sizeof(float), s);
        return e == hipSuccess ? LOCOV_OK : set_error(LOCOV_ERR_LAUNCH, "locov_winograd_wgrad_f32: memset failed");
    }
    LOCOV_REQUIRE(x && g && workspace, "locov_winograd_wgrad_f32: null pointer");
    LOCOV_REQUIRE(Cin % 4 == 0 && N % 4 == 0, "locov_winograd_wgrad_f32: Cin and N must be multiples of 4 (got %d, %d)", Cin, N);
    LOCOV_REQUIRE(((uintptr_t)x | (uintptr_t)g | (uintptr_t)dw | (uintptr_t)workspace) % 16 == 0, "locov_winograd_wgrad_f32: misaligned pointer");
    LOCOV_REQUIRE(!(flags & ~(unsigned)LOCOV_WINO_IN_ROI_MAJOR), "locov_winograd_wgrad_f32: unsupported flags 0x%x", flags);
    LOCOV_REQUIRE(workspace_bytes >= locov_winograd_wgrad_workspace_bytes(R, Cin, N), "locov_winograd_wgrad_f32: workspace too small (%lld bytes)",
                  (long long)workspace_bytes);
    float *V = static_cast<float *>(workspace);
    float *dM = V + (int64_t)NF * NF * R * Cin;
    float *dU = dM + (int64_t)NF * NF * R * N;
    float *tn_ws = dU + (int64_t)NF * NF * N * Cin;
    const bool rm = (flags & LOCOV_WINO_IN_ROI_MAJOR) != 0;
    const int64_t ld_pos = rm ? 1 : R, ld_roi = rm ? 49 : 1;
    int rc = LOCOV_OK;
    if (v_split == nullptr) {
        hipLaunchKernelGGL(wino_input_kernel<false>, dim3((unsigned)ceil_div(R * (Cin / 2), 256)), dim3(256), 0, s, x, ld_pos, ld_roi, R, Cin, V);
        rc = check_launch("locov_winograd_wgrad_f32 (input transform)");
        if (rc) return rc;
    } else
        LOCOV_REQUIRE((uintptr_t)v_split % 16 == 0, "locov_winograd_wgrad_f32_split_v: misaligned transformed input");
    float *dm_slot = split ? reinterpret_cast<float *>(static_cast<char *>(workspace) + workspace_bytes - 16) : nullptr;
    if (dm_slot && hipMemsetAsync(dm_slot, 0, 16, s) != hipSuccess) return set_error(LOCOV_ERR_LAUNCH, "locov_winograd_wgrad_f32_split: memset failed");
    hipLaunchKernelGGL(wino_input_kernel<true>, dim3((unsigned)ceil_div(R * (N / 2), 256)), dim3(256), 0, s, g, ld_pos, ld_roi, R, N, dM, 1.f,
                       static_cast<unsigned *>(nullptr), dm_slot);
    rc = check_launch("locov_winograd_wgrad_f32 (gradient transform)");
    if (rc) return rc;
    // dU_f [N, Cin] = dM_f^T . V_f   (121 problems, contraction over the ROIs)
    const int64_t tn_bytes = workspace_bytes - (int64_t)((char *)tn_ws - (char *)workspace) - 16;
    if (split) {
        // operand scales: dM from its max |.| on the device (last 16 bytes of the workspace), V as in the forward (0.25)
        // operand scales: dM from the max |.| its transform folded into the slot, V as in the forward (0.25)
        float *sc = dm_slot;
        rc = launch_gemm_tn_split(dM, (int64_t)N, R * N, v_split ? v_split : V, (int64_t)Cin, R * Cin, dU, (int64_t)Cin, (int64_t)N * Cin, R, N, Cin,
                                  NF * NF, nullptr, sc, 0.25f, overflow, tn_ws, tn_bytes, s, "locov_winograd_wgrad_f32_split (batched TN GEMM)",
                                  v_split != nullptr);
    } else
        rc = launch_gemm_tn(dM, (int64_t)N, R * N, V, (int64_t)Cin, R * Cin, dU, (int64_t)Cin, (int64_t)N * Cin, R, N, Cin, NF * NF, nullptr,
                            tn_ws, tn_bytes, s, "locov_winograd_wgrad_f32 (batched TN GEMM)");
    if (rc) return rc;
    const int64_t NC = (int64_t)N * Cin;
    hipLaunchKernelGGL(wino_unpack_wgrad_kernel, dim3((unsigned)ceil_div(NC, 256)), dim3(256), 0, s, dU, NC, Cin, row_scale, dw);
    return check_launch("locov_winograd_wgrad_f32 (filter transform)");
}

int locov_gemm_nt_batched_f32(const float *x, int64_t lda, int64_t stride_x, const float *W, int64_t stride_w, float *y,
                              int64_t ldc, int64_t stride_y, int64_t M, int N, int K, int batch, locov_stream_t stream)
{
    LOCOV_REQUIRE(M >= 0 && N > 0 && K > 0 && batch > 0, "locov_gemm_nt_batched_f32: bad shape");
    if (M == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && W && y, "locov_gemm_nt_batched_f32: null pointer");
    LOCOV_REQUIRE(K % 4 == 0 && lda % 4 == 0 && stride_x % 4 == 0 && stride_w % 4 == 0,
                  "locov_gemm_nt_batched_f32: K, lda and the operand strides must be multiples of 4");
    LOCOV_REQUIRE(lda >= K && ldc >= N, "locov_gemm_nt_batched_f32: lda < K or ldc < N");
    LOCOV_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)W % 16 == 0, "locov_gemm_nt_batched_f32: misaligned pointer");
    Epilogue epi{nullptr, nullptr, nullptr, 0u};
    return launch_gemm_nt<float, float>(x, lda, W, (int64_t)K, y, ldc, M, N, K, epi, as_stream(stream),
                                        "locov_gemm_nt_batched_f32", ConvGeom{0, 0, 0, 0, 0},
                                        Batch{batch, stride_x, stride_w, stride_y});
}

}  // extern "C"
