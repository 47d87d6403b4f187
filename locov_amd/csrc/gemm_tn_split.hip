// TN GEMM in split-operand arithmetic:  out[N,K] = row_scale[n] * sum_m a[m,n] * b[m,k]  with every fp32 product formed
// from (hi, lo) f16 pairs on the f16 matrix pipe (the arithmetic of gemm_split.hip, the shape of gemm_tn.hip): the weight
// gradients of the Res5 convolutions in the training step (roi_emb_heads.py:323,343-347 under autograd), 3 f16 MFMAs per
// 16x16x32 block instead of 16 f32 ones.
//
//   s_a a = hi + lo,  s_b b = hi + lo;   a.b ~= (hi_a.hi_b + hi_a.lo_b + lo_a.hi_b) / (s_a s_b),   fp32 accumulate.
//   s_a comes from device memory (a is a GRADIENT: its range is only known on the device, locov_split_scale_from_amax),
//   s_b is a launch parameter (b is an activation: 16, or 0.25 for Winograd-domain data, as in the forward).
//
// Both operands are row-major [M, .] matrices and the contraction runs over M, so both have to be turned K-major on their
// way into LDS.  A thread of the staging half that owns an operand loads an 8 (m) x 4 (columns) block -- eight 16-byte
// buffer loads, 512 contiguous bytes per row and half-wave -- converts it with v_fma_mix{lo,hi}_f16 pairing CONSECUTIVE m
// of one column, and writes per column one 16-byte chunk of 8 hi halves and one of 8 lo halves: the transposition is the
// register naming.  In LDS an operand tile is 128 rows (output index) x 128 B (32 m: per group of 8 m, hi then lo), the
// layout and XOR swizzle of gemm_split.hip's W tile, so the fragment reads (lane l: row l%16, m-group l/16) are its
// conflict-free ones.  128x128 output tile, 4 waves x 64x64 = 4x4 blocks, two LDS stages (64 KB), two workgroups per CU;
// M is cut into chunks as in gemm_tn.hip and the partial tiles are reduced in a fixed order by gemm_tn_reduce_kernel.
//
// B_SPLIT: b is ALREADY in the split layout of locov_split_f16x2_pack at scale b_scale (the Winograd-domain input V the forward's
// transform wrote for its own GEMMs: the weight gradient of a 3x3 convolution re-uses it instead of transforming the activation a
// second time).  The waves that stage b then load the hi and lo halves of their 8 (m) x 4 (columns) block directly (two 8-byte
// loads per row) and only transpose -- 32 v_perm_b32 instead of 64 v_fma_mix + 32 v_max -- writing the same LDS bytes the
// converting path would have produced from the fp32 values: the result is bit-identical.
#include "gemm_nt.h"

namespace locov {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int SBM = 128, SBN = 128, SBK = 32, SNT = 256;
constexpr int SROWB = 128;                         // bytes per LDS row (32 m: 4 groups x (8 hi + 8 lo) halves)
constexpr int SSTAGEB = (SBM + SBN) * SROWB;       // 32 KB per stage

__device__ __forceinline__ int swz(int row) { return (int)((0x75642031u >> (4 * ((row >> 1) & 7))) & 7u); }

__device__ __forceinline__ int tns_xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// two fp32 values (consecutive m of one column) -> packed hi pair and packed lo pair of s*x
__device__ __forceinline__ void split2(float x0, float x1, float s, unsigned &hi, unsigned &lo)
{
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
    hi = h;
    lo = l;
}

}  // namespace

template <bool B_SPLIT>
__global__ __launch_bounds__(SNT, 2) void gemm_tn_split_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                                                               int64_t ldb, float *__restrict__ P, int64_t M, int N, int K,
                                                               int splits, int64_t m_chunk, int64_t sa, int64_t sb,
                                                               const float *__restrict__ a_scale_dev, float b_scale,
                                                               unsigned *overflow)
{
    __shared__ u32x4 lds[2 * SSTAGEB / 16];
    char *const ldsb = reinterpret_cast<char *>(lds);

    const int tiles_n = (N + SBM - 1) / SBM, tiles_k = (K + SBN - 1) / SBN, tiles = tiles_n * tiles_k;
    int wg = tns_xcd_remap(blockIdx.x, gridDim.x);
    const int bs = wg / tiles;
    wg -= bs * tiles;
    const int b = bs / splits, s = bs - b * splits;
    const int n0 = (wg / tiles_k) * SBM, k0 = (wg % tiles_k) * SBN;
    const int64_t m_lo = (int64_t)s * m_chunk;
    const int64_t rows = M - m_lo < m_chunk ? M - m_lo : m_chunk;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    float a_scale, a_inv;
    split_scale_of(a_scale_dev, a_scale, a_inv);
    const float out_scale = a_inv / b_scale;

    // staging: threads 0-127 own operand A (columns n0..), 128-255 operand B (columns k0..); within a half, t % 32 = the
    // group of 4 columns, t / 32 = the group of 8 m
    // (waves 0-1 stage A, waves 2-3 stage B: said through readfirstlane so that the operand's base pointer, pitch and buffer
    //  descriptor live in scalar registers -- with a per-thread `tid < 128` every buffer load became a 12-instruction waterfall loop)
    const bool is_a = __builtin_amdgcn_readfirstlane(tid >> 6) < 2;
    const int t = tid & 127, cg = t & 31, mg = t >> 5;
    const int ld = (int)(is_a ? lda : ldb);
    const int ncol = is_a ? N : K, c0 = is_a ? n0 : k0;
    int col = c0 + cg * 4;
    col = col + 4 <= ncol ? col : ncol - 4;                            // clamped columns only feed outputs that are never stored
    const char *base = reinterpret_cast<const char *>((is_a ? A + b * sa : B + b * sb) + m_lo * (int64_t)ld);
    int64_t left = rows * (int64_t)ld * 4;                             // bytes from the running base to the chunk's end
    // (B_SPLIT, the b half: the 4 columns' hi halves sit at byte (col / 8) * 32 + (col % 8) * 2 of the row, their lo halves 16 bytes on)
    const bool pre = B_SPLIT && !is_a;
    const unsigned voff = (unsigned)((int64_t)(mg * 8) * ld * 4 + (pre ? (col >> 3) * 32 + (col & 4) * 2 : col * 4));
    const unsigned rstep = (unsigned)ld * 4u;
    const float scale = is_a ? a_scale : b_scale;
    // LDS destination of column j (row cg*4 + j of this operand's tile), chunk 2*mg (hi) / 2*mg + 1 (lo), XOR-swizzled
    int dst[4][2];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = cg * 4 + j;
#pragma unroll
        for (int hl = 0; hl < 2; hl++) dst[j][hl] = (is_a ? 0 : SBM * SROWB) + row * SROWB + (((2 * mg + hl) ^ swz(row)) * 16);
    }
    f32x4 r[8];
    float amax = 0.f;
    auto load = [&]() {
        const unsigned nrec = left > 0 ? (unsigned)left : 0u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, nrec, 0x00020000);
        if (B_SPLIT && pre) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const u32x2 h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + i * rstep, 0, 0));
                const u32x2 l = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + i * rstep + 16, 0, 0));
                r[i] = __builtin_bit_cast(f32x4, u32x4{h[0], h[1], l[0], l[1]});
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + i * rstep, 0, 0));
        }
        base += (int64_t)SBK * ld * 4;
        left -= (int64_t)SBK * ld * 4;
    };
    // B_SPLIT, the b half: r[m] = {hi halves of columns 0-1, 2-3, lo halves of columns 0-1, 2-3} of row m; column j's chunk pairs
    // the halves of consecutive m (v_perm_b32: selector bytes 0-3 address the second source, 4-7 the first)
    auto store_pre = [&](int stage) {
        char *d = ldsb + stage * SSTAGEB;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32x4 hi, lo;
            const unsigned sel = (j & 1) ? 0x07060302u : 0x05040100u;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const u32x4 e = __builtin_bit_cast(u32x4, r[2 * p]), o = __builtin_bit_cast(u32x4, r[2 * p + 1]);
                hi[p] = __builtin_amdgcn_perm(o[j >> 1], e[j >> 1], sel);
                lo[p] = __builtin_amdgcn_perm(o[2 + (j >> 1)], e[2 + (j >> 1)], sel);
            }
            *reinterpret_cast<u32x4 *>(d + dst[j][0]) = hi;
            *reinterpret_cast<u32x4 *>(d + dst[j][1]) = lo;
        }
    };
    auto store = [&](int stage) {
        char *d = ldsb + stage * SSTAGEB;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32x4 hi, lo;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const float x0 = r[2 * p][j], x1 = r[2 * p + 1][j];
                amax = fmaxf(fmaxf(amax, fabsf(x0)), fabsf(x1));
                unsigned h, l;
                split2(x0, x1, scale, h, l);
                hi[p] = h;
                lo[p] = l;
            }
            *reinterpret_cast<u32x4 *>(d + dst[j][0]) = hi;
            *reinterpret_cast<u32x4 *>(d + dst[j][1]) = lo;
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int l16 = lane & 15, kg = lane >> 4;
    int fo[2];
#pragma unroll
    for (int hl = 0; hl < 2; hl++) fo[hl] = l16 * SROWB + (((2 * kg + hl) ^ swz(l16)) * 16);
    auto compute = [&](int stage) {
        const char *As = ldsb + stage * SSTAGEB + wm * SROWB;
        const char *Bs = ldsb + stage * SSTAGEB + SBM * SROWB + wn * SROWB;
        f16x8 fa[4][2], fb[4][2];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            fa[i][0] = *reinterpret_cast<const f16x8 *>(As + i * 16 * SROWB + fo[0]);
            fa[i][1] = *reinterpret_cast<const f16x8 *>(As + i * 16 * SROWB + fo[1]);
            fb[i][0] = *reinterpret_cast<const f16x8 *>(Bs + i * 16 * SROWB + fo[0]);
            fb[i][1] = *reinterpret_cast<const f16x8 *>(Bs + i * 16 * SROWB + fo[1]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
            }
    };

    const int steps = (int)((rows + SBK - 1) / SBK);
    load();
    if (pre)
        store_pre(0);
    else
        store(0);
    if (steps > 1) load();
    __syncthreads();
#ifndef LOCOV_TNS_INTERLEAVE
#define LOCOV_TNS_INTERLEAVE 1
#endif
    int tt = 0;
    // steady state: the K-tile's 48 MFMAs and the conversion of the NEXT tile (64 v_fma_mix + 32 max, then 8 LDS writes) in ONE
    // scheduling region, interleaved 1 : 2 -- the conversion then runs on the vector ALU while the wave's own MFMAs occupy the
    // matrix pipe, instead of behind them
    if (pre) {
        // (B_SPLIT, the waves that stage b: the same loop in its own scheduling region -- 32 v_perm_b32 under the 48 MFMAs)
        for (; tt + 2 < steps; tt++) {
            const int st = tt & 1;
            compute(st);
            store_pre(st ^ 1);
            if (LOCOV_TNS_INTERLEAVE) {
#pragma unroll
                for (int i = 0; i < 16; i++) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
                for (int i = 0; i < 48; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
            }
            load();
            __syncthreads();
        }
        for (; tt < steps; tt++) {
            const int st = tt & 1;
            compute(st);
            if (tt + 1 < steps) store_pre(st ^ 1);
            __syncthreads();
        }
    } else {
    for (; tt + 2 < steps; tt++) {
        const int st = tt & 1;
        compute(st);
        store(st ^ 1);
        if (LOCOV_TNS_INTERLEAVE) {
#pragma unroll
            for (int i = 0; i < 16; i++) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // the 16 fragment reads first
#pragma unroll
            for (int i = 0; i < 48; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                 // 3 VALU
            }
        }
        load();
        __syncthreads();
    }
    for (; tt < steps; tt++) {
        const int st = tt & 1;
        compute(st);
        if (tt + 1 < steps) store(st ^ 1);
        __syncthreads();
    }
    }
    if (overflow != nullptr && amax * scale >= 65504.f) atomicOr(overflow, 1u);

    // partial tile.  C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
    float *out = P + (int64_t)bs * N * K;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int kc = k0 + wn + j * 16 + l16;
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                const int n = n0 + wm + i * 16 + 4 * kg + rg;
                if (n < N && kc < K) out[(int64_t)n * K + kc] = acc[i][j][rg] * out_scale;
            }
        }
}

// (gemm_tn.hip)
void tn_split_plan(int64_t M, int N, int K, int batch, int *splits, int64_t *m_chunk);
int launch_tn_reduce(const float *ws, int N, int K, int splits, int batch, const float *row_scale, float *out, int64_t ldo, int64_t so,
                     hipStream_t s, const char *what);

int launch_gemm_tn_split(const float *A, int64_t lda, int64_t sa, const float *B, int64_t ldb, int64_t sb, float *out, int64_t ldo,
                         int64_t so, int64_t M, int N, int K, int batch, const float *row_scale, const float *a_scale_dev, float b_scale,
                         unsigned *overflow, float *ws, int64_t ws_bytes, hipStream_t s, const char *what, bool b_split)
{
    int splits;
    int64_t m_chunk;
    tn_split_plan(M, N, K, batch, &splits, &m_chunk);
    const int64_t need = (int64_t)batch * splits * N * K * 4;
    if (ws_bytes < need) return set_error(LOCOV_ERR_INVALID_ARG, "%s: workspace too small (%lld < %lld bytes)", what, (long long)ws_bytes, (long long)need);
    if (N % 4 || K % 4 || lda % 4 || ldb % 4 || ldo % 4 || so % 4 || sa % 4 || sb % 4 || N < 4 || K < 4)
        return set_error(LOCOV_ERR_UNSUPPORTED, "%s: N, K, the row pitches and the batch strides must be multiples of 4", what);
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)out | (uintptr_t)ws) % 16)
        return set_error(LOCOV_ERR_UNSUPPORTED, "%s: pointers must be 16-byte aligned", what);
    if (!a_scale_dev || !(b_scale > 0.f)) return set_error(LOCOV_ERR_INVALID_ARG, "%s: operand scales missing", what);
    if ((m_chunk + SBK) * (lda > ldb ? lda : ldb) * 4 > 0x7fffffffLL)
        return set_error(LOCOV_ERR_INVALID_ARG, "%s: row pitch too large for 32-bit chunk offsets", what);
    const int64_t wgs = ceil_div(N, SBM) * ceil_div(K, SBN) * splits * batch;
    if (wgs > 0x7fffffffLL) return set_error(LOCOV_ERR_INVALID_ARG, "%s: problem too large", what);
    // ONE chunk and a dense, unscaled output (the 121 transform-domain problems of the Winograd weight gradient: 1 936 tiles, M = the
    // proposals): the "partial" tiles ARE the result -- written straight to `out`, no reduction pass (it was a 127 MB copy per launch)
    const bool direct = splits == 1 && row_scale == nullptr && ldo == K && (batch == 1 || so == (int64_t)N * K);
    const int trec = timing_begin(s, 7, 2.0 * (double)M * N * K * batch);        // class 7: split-operand TN GEMM
    if (b_split) {
        // (the split layout is written in groups of 8 columns; a chunk of M starts at a row, so any m_chunk is fine)
        if (K % 8 || ldb % 8 || sb % 8) return set_error(LOCOV_ERR_UNSUPPORTED, "%s: a pre-split b needs K, ldb and its batch stride to be multiples of 8", what);
        hipLaunchKernelGGL(gemm_tn_split_kernel<true>, dim3((unsigned)wgs), dim3(SNT), 0, s, A, lda, B, ldb, direct ? out : ws, M, N, K, splits, m_chunk, sa,
                           sb, a_scale_dev, b_scale, overflow);
    } else
        hipLaunchKernelGGL(gemm_tn_split_kernel<false>, dim3((unsigned)wgs), dim3(SNT), 0, s, A, lda, B, ldb, direct ? out : ws, M, N, K, splits, m_chunk, sa,
                           sb, a_scale_dev, b_scale, overflow);
    timing_end(trec, s);
    int rc = check_launch(what);
    if (rc || direct) return rc;
    return launch_tn_reduce(ws, N, K, splits, batch, row_scale, out, ldo, so, s, what);
}

}  // namespace locov

using namespace locov;

extern "C" {

int locov_gemm_tn_f32_split(const float *a, int64_t lda, int64_t stride_a, const float *b, int64_t ldb, int64_t stride_b, float *out,
                            int64_t ldo, int64_t stride_o, int64_t M, int N, int K, int batch, const float *row_scale,
                            const float *a_scale_dev, float b_scale, unsigned *overflow, void *workspace, int64_t workspace_bytes,
                            locov_stream_t stream)
{
    LOCOV_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "locov_gemm_tn_f32_split: bad shape M=%lld N=%d K=%d batch=%d", (long long)M, N, K, batch);
    LOCOV_REQUIRE(a && b && out && workspace && a_scale_dev, "locov_gemm_tn_f32_split: null pointer");
    LOCOV_REQUIRE(lda >= N && ldb >= K && ldo >= K, "locov_gemm_tn_f32_split: lda < N, ldb < K or ldo < K");
    return launch_gemm_tn_split(a, lda, stride_a, b, ldb, stride_b, out, ldo, stride_o, M, N, K, batch, row_scale, a_scale_dev, b_scale,
                                overflow, static_cast<float *>(workspace), workspace_bytes, as_stream(stream), "locov_gemm_tn_f32_split", false);
}

int locov_gemm_tn_f32_split_b(const float *a, int64_t lda, int64_t stride_a, const float *b_split, int64_t ldb, int64_t stride_b, float *out,
                              int64_t ldo, int64_t stride_o, int64_t M, int N, int K, int batch, const float *row_scale,
                              const float *a_scale_dev, float b_scale, unsigned *overflow, void *workspace, int64_t workspace_bytes,
                              locov_stream_t stream)
{
    LOCOV_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "locov_gemm_tn_f32_split_b: bad shape M=%lld N=%d K=%d batch=%d", (long long)M, N, K, batch);
    LOCOV_REQUIRE(a && b_split && out && workspace && a_scale_dev, "locov_gemm_tn_f32_split_b: null pointer");
    LOCOV_REQUIRE(lda >= N && ldb >= K && ldo >= K, "locov_gemm_tn_f32_split_b: lda < N, ldb < K or ldo < K");
    return launch_gemm_tn_split(a, lda, stride_a, b_split, ldb, stride_b, out, ldo, stride_o, M, N, K, batch, row_scale, a_scale_dev, b_scale,
                                overflow, static_cast<float *>(workspace), workspace_bytes, as_stream(stream), "locov_gemm_tn_f32_split_b", true);
}

}  // extern "C"
