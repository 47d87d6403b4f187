// TN GEMM  out[N,K] = row_scale[n] * sum_m a[m,n] * b[m,k]  on the gfx950 f32 matrix pipe -- the weight gradients of the
// Res5 convolutions when the LSM head trains (roi_emb_heads.py:323,343-347 under autograd; the reference gets them from
// cuDNN's wgrad): both operands are row-major "pixel rows x channels" matrices (the activations and the output
// gradients as every other kernel of this library lays them out), the contraction runs over the LONG dimension M
// (49 x ROIs rows), the result is a weight-sized matrix.
//
//   * 128x128 output tile, 4 waves x 64x64 as 2x2 v_mfma_f32_32x32x2_f32 accumulators (exact fp32 products).
//   * M is cut into `splits` chunks so that a weight-sized output (64 tiles for [2048,512]) still fills 256 CUs; every
//     (tile, chunk) workgroup writes its partial tile to the workspace and gemm_tn_reduce_kernel adds the chunks in a
//     fixed order (deterministic; no atomics) and applies the FrozenBN scale of the output channel.
//   * Staging: a K-step is 32 rows of m: 32 x 128 floats per operand, 16-byte buffer loads (rows past the chunk fall
//     outside the descriptor's num_records and read as zero -> ragged M needs no masks), written to LDS as they come
//     ([m][128 + 32 pad] floats).  The MFMA operand of lane l is (row l%32 of the tile, contraction index l/32): one
//     ds_read_b32 at m*PITCH + column -- the transposition costs nothing, lanes 0-31 and 32-63 sit in disjoint bank halves.
//   * batch > 1: independent problems of one shape in one launch (the 121 transform-domain problems of the Winograd
//     wgrad, winograd.hip).
#include "gemm_nt.h"

#include <cstdlib>

namespace locov {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TBM = 128, TBN = 128, TBK = 32, TNT = 256;
constexpr int PITCH = TBM + 32;                 // floats per LDS row
constexpr int TSTAGE = 2 * TBK * PITCH;         // floats per stage (A rows, then B rows)

__device__ __forceinline__ int tn_xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

}  // namespace

// partial[(b*splits + s)][n][k] = sum_{m in chunk s} a_b[m,n] * b_b[m,k]
__global__ __launch_bounds__(TNT, 2) void gemm_tn_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                                                         int64_t ldb, float *__restrict__ P, int64_t M, int N, int K,
                                                         int splits, int64_t m_chunk, int64_t sa, int64_t sb)
{
    __shared__ f32x4 lds4[2 * TSTAGE / 4];
    float *const lds = reinterpret_cast<float *>(lds4);

    const int tiles_n = (N + TBM - 1) / TBM, tiles_k = (K + TBN - 1) / TBN, tiles = tiles_n * tiles_k;
    int wg = tn_xcd_remap(blockIdx.x, gridDim.x);      // consecutive ids (one XCD's run) = the tiles of one chunk: they share its rows in L2
    const int bs = wg / tiles;                         // b * splits + s
    wg -= bs * tiles;
    const int b = bs / splits, s = bs - b * splits;
    const int n0 = (wg / tiles_k) * TBM, k0 = (wg % tiles_k) * TBN;
    const int64_t m_lo = (int64_t)s * m_chunk;
    const int64_t rows = M - m_lo < m_chunk ? M - m_lo : m_chunk;      // > 0 by construction
    const char *a_base = reinterpret_cast<const char *>(A + b * sa + m_lo * lda);
    const char *b_base = reinterpret_cast<const char *>(B + b * sb + m_lo * ldb);
    int64_t a_left = rows * lda * 4, b_left = rows * ldb * 4;          // bytes from the running base to the chunk's end

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    // staging: chunk i of a thread = row (tid + i*256) / 32 of the K-step, 16-byte chunk (tid + i*256) % 32 of the tile row;
    // columns past N / K are clamped to an in-bounds chunk (they only feed outputs that are never stored)
    unsigned a_off[4], b_off[4];
    int s_off[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int idx = tid + i * TNT, row = idx >> 5, ch = idx & 31;
        int ca = n0 + ch * 4, cb = k0 + ch * 4;
        ca = ca + 4 <= N ? ca : N - 4;
        cb = cb + 4 <= K ? cb : K - 4;
        a_off[i] = (unsigned)(((int64_t)row * lda + ca) * 4);
        b_off[i] = (unsigned)(((int64_t)row * ldb + cb) * 4);
        s_off[i] = row * PITCH + ch * 4;
    }
    f32x4 ra[4], rb[4];
    auto load = [&]() {
        const unsigned na = a_left > 0 ? (unsigned)a_left : 0u, nb = b_left > 0 ? (unsigned)b_left : 0u;
        const __amdgpu_buffer_rsrc_t r_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a_base), 0, na, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(b_base), 0, nb, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; i++) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_a, a_off[i], 0, 0));
#pragma unroll
        for (int i = 0; i < 4; i++) rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_b, b_off[i], 0, 0));
        a_base += (int64_t)TBK * lda * 4;
        b_base += (int64_t)TBK * ldb * 4;
        a_left -= (int64_t)TBK * lda * 4;
        b_left -= (int64_t)TBK * ldb * 4;
    };
    auto store = [&](int stage) {
        float *As = lds + stage * TSTAGE, *Bs = As + TBK * PITCH;
#pragma unroll
        for (int i = 0; i < 4; i++) *reinterpret_cast<f32x4 *>(As + s_off[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; i++) *reinterpret_cast<f32x4 *>(Bs + s_off[i]) = rb[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = f32x16{};

    const int fr = lane & 31, fh = lane >> 5;
    auto compute = [&](int stage) {
        const float *As = lds + stage * TSTAGE + fh * PITCH + wm + fr;
        const float *Bs = lds + stage * TSTAGE + TBK * PITCH + fh * PITCH + wn + fr;
#pragma unroll
        for (int kk = 0; kk < TBK / 2; kk++) {
            const float a0 = As[2 * kk * PITCH], a1 = As[2 * kk * PITCH + 32];
            const float b0 = Bs[2 * kk * PITCH], b1 = Bs[2 * kk * PITCH + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    };

    const int steps = (int)((rows + TBK - 1) / TBK);
    load();
    store(0);
    if (steps > 1) load();
    __syncthreads();
    for (int t = 0; t < steps; t++) {
        const int st = t & 1;
        compute(st);
        if (t + 1 < steps) {
            store(st ^ 1);                 // K-step t+1 (its loads were issued a whole step ago)
            if (t + 2 < steps) load();
        }
        __syncthreads();
    }

    // partial tile: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float *out = P + (int64_t)bs * N * K;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int kc = k0 + wn + j * 32 + fr;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = n0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (n < N && kc < K) out[(int64_t)n * K + kc] = acc[i][j][r];
            }
        }
}

// out[b][n][k] = row_scale[n] * sum_s partial[b*splits + s][n][k]      (fixed summation order)
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float *__restrict__ P, int N, int K, int splits, int batch,
                                                             const float *__restrict__ row_scale, float *__restrict__ out,
                                                             int64_t ldo, int64_t so)
{
    const int k4 = K >> 2;
    const int64_t per = (int64_t)N * k4, total = per * batch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / per, rem = i - b * per;
        const int n = (int)(rem / k4), k = (int)(rem - (int64_t)n * k4) * 4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        const float *p = P + (b * splits) * (int64_t)N * K + (int64_t)n * K + k;
        for (int s = 0; s < splits; s++) a += *reinterpret_cast<const f32x4 *>(p + (int64_t)s * N * K);
        if (row_scale) a *= row_scale[n];
        *reinterpret_cast<f32x4 *>(out + b * so + (int64_t)n * ldo + k) = a;
    }
}

// chunking of M: enough (tile, chunk) workgroups to fill the chip, chunks of at least 256 rows, multiples of TBK
// (shared with gemm_tn_split.hip)
void tn_split_plan(int64_t M, int N, int K, int batch, int *splits, int64_t *m_chunk)
{
    const int64_t tiles = ceil_div(N, TBM) * ceil_div(K, TBN) * batch;
    static const int64_t target = [] { const char *e = getenv("LOCOV_TN_WGS"); return e && atoll(e) > 0 ? (int64_t)atoll(e) : (int64_t)1024; }();
    int64_t s = ceil_div(target, tiles);
    const int64_t smax = ceil_div(M, 256);
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    int64_t mc = ceil_div(ceil_div(M, s), TBK) * TBK;
    *m_chunk = mc;
    *splits = (int)ceil_div(M, mc);
}

int launch_tn_reduce(const float *ws, int N, int K, int splits, int batch, const float *row_scale, float *out, int64_t ldo, int64_t so,
                     hipStream_t s, const char *what)
{
    const int64_t total = (int64_t)batch * N * (K / 4);
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 8192 ? ceil_div(total, 256) : 8192);
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, N, K, splits, batch, row_scale, out, ldo, so);
    return check_launch(what);
}

int launch_gemm_tn(const float *A, int64_t lda, int64_t sa, const float *B, int64_t ldb, int64_t sb, float *out, int64_t ldo,
                   int64_t so, int64_t M, int N, int K, int batch, const float *row_scale, float *ws, int64_t ws_bytes,
                   hipStream_t s, const char *what)
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
    if ((m_chunk + TBK) * (lda > ldb ? lda : ldb) * 4 > 0x7fffffffLL)
        return set_error(LOCOV_ERR_INVALID_ARG, "%s: row pitch too large for 32-bit chunk offsets", what);
    const int64_t wgs = ceil_div(N, TBM) * ceil_div(K, TBN) * splits * batch;
    if (wgs > 0x7fffffffLL) return set_error(LOCOV_ERR_INVALID_ARG, "%s: problem too large", what);
    // (one chunk, dense unscaled output: the partial tiles are the result, see gemm_tn_split.hip)
    const bool direct = splits == 1 && row_scale == nullptr && ldo == K && (batch == 1 || so == (int64_t)N * K);
    const int trec = timing_begin(s, 6, 2.0 * (double)M * N * K * batch);        // class 6: TN (wgrad) GEMM
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)wgs), dim3(TNT), 0, s, A, lda, B, ldb, direct ? out : ws, M, N, K, splits, m_chunk, sa, sb);
    timing_end(trec, s);
    int rc = check_launch(what);
    if (rc || direct) return rc;
    return launch_tn_reduce(ws, N, K, splits, batch, row_scale, out, ldo, so, s, what);
}

int64_t gemm_tn_workspace_bytes(int64_t M, int N, int K, int batch)
{
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 0;
    int splits;
    int64_t m_chunk;
    tn_split_plan(M, N, K, batch, &splits, &m_chunk);
    return (int64_t)batch * splits * N * K * 4;
}

}  // namespace locov

using namespace locov;

extern "C" {

int64_t locov_gemm_tn_workspace_bytes(int64_t M, int N, int K, int batch) { return gemm_tn_workspace_bytes(M, N, K, batch); }

int locov_gemm_tn_f32(const float *a, int64_t lda, int64_t stride_a, const float *b, int64_t ldb, int64_t stride_b, float *out,
                      int64_t ldo, int64_t stride_o, int64_t M, int N, int K, int batch, const float *row_scale, void *workspace,
                      int64_t workspace_bytes, locov_stream_t stream)
{
    LOCOV_REQUIRE(M >= 0 && N > 0 && K > 0 && batch > 0, "locov_gemm_tn_f32: bad shape M=%lld N=%d K=%d batch=%d", (long long)M, N, K, batch);
    LOCOV_REQUIRE(out, "locov_gemm_tn_f32: null output");
    LOCOV_REQUIRE(lda >= N && ldb >= K && ldo >= K, "locov_gemm_tn_f32: lda < N, ldb < K or ldo < K");
    if (M == 0) {
        for (int bi = 0; bi < batch; bi++) {
            hipError_t e = hipMemset2DAsync(out + bi * stride_o, (size_t)ldo * 4, 0, (size_t)K * 4, (size_t)N, as_stream(stream));
            if (e != hipSuccess) return set_error(LOCOV_ERR_LAUNCH, "locov_gemm_tn_f32: memset failed: %s", hipGetErrorString(e));
        }
        return LOCOV_OK;
    }
    LOCOV_REQUIRE(a && b && workspace, "locov_gemm_tn_f32: null pointer");
    return launch_gemm_tn(a, lda, stride_a, b, ldb, stride_b, out, ldo, stride_o, M, N, K, batch, row_scale,
                          static_cast<float *>(workspace), workspace_bytes, as_stream(stream), "locov_gemm_tn_f32");
}

}  // extern "C"
