// Internal interface of gemm_nt.hip (shared with head.hip).
#pragma once
#include "common.h"

namespace locov {

struct Epilogue {
    const float *scale;     // [N] or null
    const float *shift;     // [N] or null  (bias)
    const float *residual;  // [M, ldc] or null
    unsigned flags;
    const float *mask;      // [M, ldc] or null: the finished value is kept where mask > 0 and zeroed elsewhere (the ReLU
                            // backward of a saved activation, fused into the data-gradient GEMMs; f32 NT kernel only)
    float *amax_out;        // split GEMM only, or null: a 16-byte operand-scale slot {s, 1/s, max bits, -}; the launch folds
                            // max |finished value| into word 2 (atomicMax on the bit pattern) -- the tensor it writes is the
                            // operand of a later split GEMM whose scale is then derived from that word (split_scale_of)
};

// Operand scale from a 16-byte slot: {s, 1/s} as written by locov_split_scale_from_amax, or -- s == 0: a zeroed slot that
// producers only folded their max |x| into -- the power of two that puts that max in [2^12, 2^13) (1 for an all-zero tensor).
__device__ __forceinline__ void split_scale_of(const float *slot, float &s, float &inv)
{
    s = slot[0];
    inv = slot[1];
    if (s == 0.f) {
        const float amax = __uint_as_float(reinterpret_cast<const unsigned *>(slot)[2]);
        s = 1.f;
        if (amax > 0.f && amax < 3.0e38f) {
            int e;
            frexpf(amax, &e);
            s = ldexpf(1.f, 13 - e);
        }
        inv = 1.f / s;
    }
}

// max |.| of a wave's values -> the slot's word 2.  Thousands of atomics on one address serialise at L2 (20 k waves of a data-gradient
// GEMM cost more than the separate reduction pass they replace), so a wave first LOOKS at the running max (a plain L2 read) and
// only a wave that would raise it issues the atomic: after the first few arrivals almost none does.
__device__ __forceinline__ void amax_fold(float *slot, float m)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) {
        unsigned *w = reinterpret_cast<unsigned *>(slot) + 2;
        const unsigned bits = __float_as_uint(m);                      // non-negative floats order like their bit patterns
        if (bits > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(w, bits);
    }
}

// H == 0: plain GEMM.  H > 0: A is the pixel matrix of independent HxW tiles with Cin channels
// and the GEMM is the implicit form of a 3x3 / pad 1 / stride 1 convolution (K = 9*Cin).
// R > 0 additionally says the rows are POSITION-major ([H*W][R] instead of [R][H*W]).
// group: ROI blocks per lock-step tile group of the position-major convolution (0 = default).
struct ConvGeom {
    int H, W, Cin, R, group;
};

// count > 1: `count` independent GEMMs of the same shape in one launch; problem b uses
// A + b*sa, B + b*sb, C + b*sc (elements).  Plain GEMM only (no conv geometry, no residual).
struct Batch {
    int count;
    int64_t sa, sb, sc;
};

// y[M,N] = epi(A[M,K] . B[N,K]^T); T = float (f32 MFMA) or __bf16 (bf16 MFMA, fp32 accumulate)
template <typename T, typename TOut>
int launch_gemm_nt(const T *A, int64_t lda, const T *B, int64_t ldb, TOut *C, int64_t ldc, int64_t M, int N, int K,
                   const Epilogue &epi, hipStream_t s, const char *what, const ConvGeom &cg = ConvGeom{0, 0, 0, 0, 0},
                   const Batch &bt = Batch{1, 0, 0, 0});

// Per-launch timing hooks of locov_gemm_timing_* (gemm_nt.hip): begin() returns a record index or -1 when timing is
// off; cls = kernel class (include/locov_hip.h), flops = what the launch executes, bytes = its ALGORITHMIC HBM bytes (every operand
// read once, every result written once; 0 = not stated).
int timing_begin(hipStream_t s, int cls, double flops, double bytes = 0.0);
void timing_end(int idx, hipStream_t s);

// y[M,N] = epi(A[M,K] . W[N,K]^T) with fp32 A split on the fly (scaled by a_scale) and W pre-split into f16 (hi, lo)
// pairs of w_scale * W (gemm_split.hip); same Epilogue / Batch meaning as launch_gemm_nt.
// overflow: optional device word that the kernel ORs 1 into when an A value left the range a_scale covers (locov_hip.h).
int launch_gemm_split(const float *A, int64_t lda, const void *Wsplit, float *C, int64_t ldc, int64_t M, int N, int K,
                      const Epilogue &epi, float a_scale, float w_scale, hipStream_t s, const char *what,
                      const Batch &bt = Batch{1, 0, 0, 0}, unsigned *overflow = nullptr, const float *a_scale_dev = nullptr);

// the same on a 256 x 256 tile (gemm_split_big.hip) for launches with BOTH operands pre-split that fill the chip with such tiles
bool gemm_split_big_applicable(int64_t lda, int64_t ldc, int64_t M, int N, int K, const Epilogue &epi, const Batch &bt,
                               const float *a_scale_dev);
int launch_gemm_split_big(const float *A, int64_t lda, const void *Wsplit, float *C, int64_t ldc, int64_t M, int N, int K,
                          const Epilogue &epi, float a_scale, float w_scale, hipStream_t s, const char *what, const Batch &bt,
                          unsigned *overflow);

bool gemm_split_big_segmean_applicable(int64_t lda, int64_t M, int N, int K, const Epilogue &epi, int seg);
int64_t gemm_split_big_segmean_workspace_bytes(int64_t M, int N);
int launch_gemm_split_big_segmean(const float *A, int64_t lda, const void *Wsplit, int64_t M, int N, int K, const Epilogue &epi, int seg,
                                  float a_scale, float w_scale, float *partial, float *out, hipStream_t s, const char *what, unsigned *overflow);

// ... with the Winograd input transform of the 3x3 convolution behind it in the epilogue: V [121][M/49][N], split layout x v_scale
bool gemm_split_big_wino_applicable(int64_t lda, int64_t M, int N, int K, const Epilogue &epi);
int launch_gemm_split_big_wino(const float *A, int64_t lda, const void *Wsplit, int64_t M, int N, int K, const Epilogue &epi, float a_scale,
                               float w_scale, float *V, float v_scale, hipStream_t s, const char *what, unsigned *overflow);

// ROIAlign (even bins of a 14 x 14 pooler, ROI-major) + per-channel affine + ReLU with the same transform behind it (roi_align_nhwc.hip)
bool roi_align_nhwc_wino_applicable(int C, int64_t R);
int launch_roi_align_nhwc_wino(const float *feat, int N, int H, int W, int C, int64_t feat_ld, const float *rois, int64_t R, int pooled,
                               float spatial_scale, int sampling_ratio, int aligned, const float *ch_scale, const float *ch_shift, int relu,
                               float *V, float v_scale, unsigned *overflow, hipStream_t s);

// out[b][N,K] = row_scale[n] * sum_m A_b[m,n] * B_b[m,k]  (gemm_tn.hip: the weight-gradient GEMM; problem b uses
// A + b*sa, B + b*sb, out + b*so; ws = gemm_tn_workspace_bytes(M, N, K, batch) bytes of device memory)
int launch_gemm_tn(const float *A, int64_t lda, int64_t sa, const float *B, int64_t ldb, int64_t sb, float *out, int64_t ldo,
                   int64_t so, int64_t M, int N, int K, int batch, const float *row_scale, float *ws, int64_t ws_bytes,
                   hipStream_t s, const char *what);
int64_t gemm_tn_workspace_bytes(int64_t M, int N, int K, int batch);
// the same in split-operand arithmetic (gemm_tn_split.hip): a = gradient, scale {s, 1/s} in device memory; b = activation, scale by value
int launch_gemm_tn_split(const float *A, int64_t lda, int64_t sa, const float *B, int64_t ldb, int64_t sb, float *out, int64_t ldo,
                         int64_t so, int64_t M, int N, int K, int batch, const float *row_scale, const float *a_scale_dev, float b_scale,
                         unsigned *overflow, float *ws, int64_t ws_bytes, hipStream_t s, const char *what, bool b_split = false);

}  // namespace locov
