// Internal interface of gemm_nt.hip (shared with head.hip / res5.hip).
#pragma once
#include "common.h"

namespace locov {

struct Epilogue {
    const float *scale;     // [N] or null
    const float *shift;     // [N] or null  (bias)
    const float *residual;  // [M, ldc] or null
    unsigned flags;
};

// y[M,N] = epi(A[M,K] . B[N,K]^T); T = float (f32 MFMA) or __bf16 (bf16 MFMA, fp32 accumulate)
template <typename T>
int launch_gemm_nt(const T *A, int64_t lda, const T *B, int64_t ldb, float *C, int64_t ldc, int64_t M, int N,
                   int K, const Epilogue &epi, hipStream_t s, const char *what);

}  // namespace locov
