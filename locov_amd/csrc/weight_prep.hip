// Every split-operand form of the Res5 convolution weights a TRAINING step needs, in ONE launch.
//
// The LSM configuration trains the Res5 convolutions (configs/coco_lsm.yaml:8), so the optimizer moves all ten weights every
// step and every derived operand of the hand-written path has to be rebuilt per step (roi_emb_heads.py:323,343-347 under
// autograd): per 1x1 convolution the forward operand W [N,K] and the data gradient's operand (s W)^T [K,N]; per 3x3
// convolution the Winograd-domain filter U = (G (x) G) w of the proposals' 7x7 tiles, the im2col filter [N, 9 Cin] of the
// whole-grid call, and both forms of the FLIPPED filter flip(s w) [Cin, N, 3, 3] for the data gradients -- each followed by
// the (hi, lo) f16 split of gemm_split.hip.  As separate launches that was 48 small kernels per step (transpose / flip / pack,
// then split_pack8 per result: 0.7 ms of a 14 ms step, and a third of its launches); their results depend on nothing but the
// weights, so one launch right behind the optimizer step produces all of them -- enqueued while the host still waits for the
// labelling, where the GPU would otherwise idle.
//
// One thread = one 32-byte run of the split layout (8 consecutive columns of one row: 8 hi halves, 8 lo halves), or, for the
// Winograd jobs, the 121 runs a group of 8 filters yields.  The arithmetic per element is that of the kernels this replaces
// (weight_transpose_scale / conv3x3_weight_flip: one fp32 multiply by the FrozenBN scale; wino_pack_weight: the filter
// transform in fp64, rounded once; split_pack8: hi = f16(s x), lo = f16(s x - hi)), so the operands are bit-identical to the
// multi-launch path, which stays for the steps that choose the power-of-two operand scales afresh (a host read of max |.|).
#include "common.h"
#include "winograd_filter.h"

namespace locov {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct PrepJob {
    const float *w;          // the convolution weight as stored: [N, K] (1x1) or [N, Cin, 3, 3]
    const float *row_scale;  // FrozenBN scale s[n] folded into the backward operands (null: none)
    u32x4 *out;              // the split-layout operand
    float scale;             // power-of-two operand scale of the split
    int kind, N, K;          // K = Cin for the 3x3 kinds
    unsigned first_block, blocks;
};

struct PrepJobs {
    PrepJob job[LOCOV_WEIGHT_PREP_MAX_JOBS];
    int count;
};

__device__ __forceinline__ void store_split8(u32x4 *dst, const float (&v)[8], float s, bool &over)
{
    _Float16 h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float x = v[e] * s;
        over |= fabsf(x) >= 65504.f;
        h[e] = (_Float16)x;
        l[e] = (_Float16)(x - (float)h[e]);
    }
    u32x4 ho, lo;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        ho[e] = (unsigned)__builtin_bit_cast(unsigned short, h[2 * e]) | ((unsigned)__builtin_bit_cast(unsigned short, h[2 * e + 1]) << 16);
        lo[e] = (unsigned)__builtin_bit_cast(unsigned short, l[2 * e]) | ((unsigned)__builtin_bit_cast(unsigned short, l[2 * e + 1]) << 16);
    }
    dst[0] = ho;
    dst[1] = lo;
}

// U[f][row][col8 ..] for one (row, group of 8 columns): g(e) = the 3x3 filter of column e, already scaled
template <typename FilterOf>
__device__ __forceinline__ void wino_runs(u32x4 *out, int64_t plane_groups, int64_t group, float s, bool &over, FilterOf filter)
{
    using wino::NF;
    double g[8][3][3];
#pragma unroll
    for (int e = 0; e < 8; e++) filter(e, g[e]);
    for (int fy = 0; fy < NF; fy++) {
        double t[8][3];
#pragma unroll
        for (int e = 0; e < 8; e++)
#pragma unroll
            for (int b = 0; b < 3; b++) t[e][b] = wino::filter_dot3(fy, g[e][0][b], g[e][1][b], g[e][2][b]);
        for (int fx = 0; fx < NF; fx++) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = (float)wino::filter_dot3(fx, t[e][0], t[e][1], t[e][2]);
            store_split8(out + 2 * ((int64_t)(fy * NF + fx) * plane_groups + group), v, s, over);
        }
    }
}

__global__ __launch_bounds__(256) void weight_prep_kernel(PrepJobs jobs, unsigned *overflow)
{
    int ji = 0;
    while (ji + 1 < jobs.count && blockIdx.x >= jobs.job[ji + 1].first_block) ji++;
    const PrepJob &jb = jobs.job[ji];
    const float *__restrict__ w = jb.w;
    const float *__restrict__ rs = jb.row_scale;
    const int N = jb.N, K = jb.K;
    const float s = jb.scale;
    bool over = false;
    const int64_t stride = (int64_t)jb.blocks * 256;
    const int64_t t0 = (int64_t)(blockIdx.x - jb.first_block) * 256 + threadIdx.x;
    switch (jb.kind) {
    case LOCOV_PREP_PLAIN: {                               // out [N, K] = w
        const int64_t total = (int64_t)N * K / 8;
        for (int64_t i = t0; i < total; i += stride) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(w + i * 8);
            const f32x4 a = src[0], b = src[1];
            const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            store_split8(jb.out + 2 * i, v, s, over);
        }
        break;
    }
    case LOCOV_PREP_TRANSPOSE: {                           // out [K, N]: out[k][n] = w[n][k] * s[n]
        const int n8 = N / 8;
        const int64_t total = (int64_t)K * n8;
        // (thread order: k fastest inside a group of 8 filters, so that a wave reads 64 consecutive k of each of its 8 rows)
        for (int64_t i = t0; i < total; i += stride) {
            const int k = (int)(i % K);
            const int g = (int)(i / K);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int n = g * 8 + e;
                v[e] = w[(int64_t)n * K + k] * (rs ? rs[n] : 1.f);
            }
            store_split8(jb.out + 2 * ((int64_t)k * n8 + g), v, s, over);
        }
        break;
    }
    case LOCOV_PREP_IM2COL: {                              // out [N, 9 Cin]: out[n][tap * Cin + c] = w[n][c][tap]
        const int c8 = K / 8;
        const int64_t total = (int64_t)N * 9 * c8;
        for (int64_t i = t0; i < total; i += stride) {
            const int g = (int)(i % c8);
            const int tap = (int)((i / c8) % 9);
            const int64_t n = i / ((int64_t)c8 * 9);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = w[(n * K + g * 8 + e) * 9 + tap];
            store_split8(jb.out + 2 * i, v, s, over);
        }
        break;
    }
    case LOCOV_PREP_IM2COL_FLIP: {                         // out [Cin, 9 N]: out[c][tap * N + n] = w[n][c][8 - tap] * s[n]
        const int n8 = N / 8;
        const int64_t total = (int64_t)K * 9 * n8;
        for (int64_t i = t0; i < total; i += stride) {
            const int g = (int)(i % n8);
            const int tap = (int)((i / n8) % 9);
            const int64_t c = i / ((int64_t)n8 * 9);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int n = g * 8 + e;
                v[e] = w[((int64_t)n * K + c) * 9 + (8 - tap)] * (rs ? rs[n] : 1.f);
            }
            store_split8(jb.out + 2 * i, v, s, over);
        }
        break;
    }
    case LOCOV_PREP_WINO: {                                // out [121, N, Cin] = (G (x) G) w[n][c]
        const int c8 = K / 8;
        const int64_t total = (int64_t)N * c8;
        for (int64_t i = t0; i < total; i += stride) {
            const float *src = w + i * 72;                 // 8 consecutive input channels of one filter: 72 contiguous floats
            wino_runs(jb.out, total, i, s, over, [&](int e, double (&g)[3][3]) {
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 3; b++) g[a][b] = (double)src[e * 9 + a * 3 + b];
            });
        }
        break;
    }
    case LOCOV_PREP_WINO_FLIP: {                           // out [121, Cin, N] = (G (x) G) flip(s w)[c][n]
        const int n8 = N / 8;
        const int64_t total = (int64_t)K * n8;
        for (int64_t i = t0; i < total; i += stride) {
            const int g8 = (int)(i % n8);
            const int64_t c = i / n8;
            wino_runs(jb.out, total, i, s, over, [&](int e, double (&g)[3][3]) {
                const int n = g8 * 8 + e;
                const float sc = rs ? rs[n] : 1.f;
                const float *src = w + ((int64_t)n * K + c) * 9;
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 3; b++) g[a][b] = (double)(src[8 - (a * 3 + b)] * sc);
            });
        }
        break;
    }
    default: break;
    }
    if (overflow != nullptr && over) atomicOr(overflow, 1u);        // a re-used scale no longer covers the data
}

}  // namespace

}  // namespace locov

using namespace locov;

extern "C" int locov_res5_weight_prep(const locov_weight_prep_job *jobs, int n_jobs, unsigned *overflow, locov_stream_t stream)
{
    LOCOV_REQUIRE(n_jobs >= 0 && n_jobs <= LOCOV_WEIGHT_PREP_MAX_JOBS, "locov_res5_weight_prep: 0..%d jobs per launch", LOCOV_WEIGHT_PREP_MAX_JOBS);
    if (n_jobs == 0) return LOCOV_OK;
    LOCOV_REQUIRE(jobs, "locov_res5_weight_prep: null job list");
    PrepJobs pj{};
    unsigned next = 0;
    for (int i = 0; i < n_jobs; i++) {
        const locov_weight_prep_job &j = jobs[i];
        LOCOV_REQUIRE(j.w && j.out && ((uintptr_t)j.w | (uintptr_t)j.out) % 16 == 0, "locov_res5_weight_prep: job %d: null or misaligned pointer", i);
        LOCOV_REQUIRE(j.N > 0 && j.K > 0 && j.scale > 0.f, "locov_res5_weight_prep: job %d: bad shape or scale", i);
        int64_t threads;
        switch (j.kind) {
        case LOCOV_PREP_PLAIN:
            LOCOV_REQUIRE(j.K % 32 == 0, "locov_res5_weight_prep: job %d: K %% 32", i);
            threads = (int64_t)j.N * j.K / 8;
            break;
        case LOCOV_PREP_TRANSPOSE:
            LOCOV_REQUIRE(j.N % 32 == 0, "locov_res5_weight_prep: job %d: N %% 32 (the transposed operand's K)", i);
            threads = (int64_t)j.K * j.N / 8;
            break;
        case LOCOV_PREP_IM2COL:
            LOCOV_REQUIRE(j.K % 32 == 0, "locov_res5_weight_prep: job %d: Cin %% 32", i);
            threads = (int64_t)j.N * 9 * j.K / 8;
            break;
        case LOCOV_PREP_IM2COL_FLIP:
            LOCOV_REQUIRE(j.N % 32 == 0, "locov_res5_weight_prep: job %d: N %% 32", i);
            threads = (int64_t)j.K * 9 * j.N / 8;
            break;
        case LOCOV_PREP_WINO:
            LOCOV_REQUIRE(j.K % 32 == 0, "locov_res5_weight_prep: job %d: Cin %% 32", i);
            threads = (int64_t)j.N * j.K / 8;
            break;
        case LOCOV_PREP_WINO_FLIP:
            LOCOV_REQUIRE(j.N % 32 == 0, "locov_res5_weight_prep: job %d: N %% 32", i);
            threads = (int64_t)j.K * j.N / 8;
            break;
        default:
            return set_error(LOCOV_ERR_INVALID_ARG, "locov_res5_weight_prep: job %d: unknown kind %d", i, j.kind);
        }
        // (the Winograd jobs write 121 runs per thread: one thread per workgroup slot; the others loop over a capped grid)
        const bool wino = j.kind == LOCOV_PREP_WINO || j.kind == LOCOV_PREP_WINO_FLIP;
        int64_t blocks = ceil_div(threads, 256);
        if (!wino && blocks > 2048) blocks = 2048;
        PrepJob &d = pj.job[i];
        d.w = j.w;
        d.row_scale = j.row_scale;
        d.out = static_cast<u32x4 *>(j.out);
        d.scale = j.scale;
        d.kind = j.kind;
        d.N = j.N;
        d.K = j.K;
        d.first_block = next;
        d.blocks = (unsigned)blocks;
        LOCOV_REQUIRE((int64_t)next + blocks < 0x7fffffffLL, "locov_res5_weight_prep: too many workgroups");
        next += (unsigned)blocks;
    }
    pj.count = n_jobs;
    hipLaunchKernelGGL(weight_prep_kernel, dim3(next), dim3(256), 0, as_stream(stream), pj, overflow);
    return check_launch("locov_res5_weight_prep");
}
