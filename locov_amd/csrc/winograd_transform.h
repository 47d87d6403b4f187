// Device pieces of the Winograd input transform shared by wino_input_kernel (winograd.hip) and the fused form in the epilogue of the
// 256x256 split GEMM (gemm_split_big.hip, MODE_WINO): ONE source for the arithmetic, so that the transform-domain tensor V is the
// same bits whichever kernel wrote it (the head's results must not depend on which launches qualify for the fused form).
#pragma once

#include "gemm_nt.h"
#include "winograd_tables.h"

// Cache policy of the transform-domain tensors (A/B: tools/ab_wino.sh, 8 000 ROIs): V is WRITTEN with the default policy
// (input transform 561 -> 531 us against non-temporal stores), M is READ non-temporally (output transform 556 us against 593)
#ifndef LOCOV_WINO_NT_STORE
#define LOCOV_WINO_NT_STORE 0
#endif
#ifndef LOCOV_WINO_NT_LOAD
#define LOCOV_WINO_NT_LOAD 1
#endif

namespace locov {

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ void wino_store(const T &v, T *p)
{
    if (LOCOV_WINO_NT_STORE)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}
template <typename T>
__device__ __forceinline__ T wino_load(const T *p)
{
    return LOCOV_WINO_NT_LOAD ? __builtin_nontemporal_load(p) : *p;
}

// Store the channel pair (c, c+1) of a row in the SPLIT layout of locov_split_f16x2_pack (per 8 channels: 8 hi halves, then 8
// lo halves of s*v; the row keeps its fp32 size): the split GEMM then stages this tensor by LDS DMA with no conversion of
// its own (gemm_split.hip, ASPLIT).  A lane owns two channels = 4 bytes of hi and 4 of lo; neighbouring lanes (c and c+2,
// same group of 8) trade one word so that the even one stores the hi halves of four channels and the odd one their lo
// halves -- 8 contiguous bytes per lane, a wave's 512-byte row segment fully written by one instruction, as in fp32.
typedef unsigned wino_u32x2 __attribute__((ext_vector_type(2)));

// the 8 bytes lane (channel pair c) stores, and their byte offset in the row
__device__ __forceinline__ wino_u32x2 split_pair_words(int c, f32x2 v, float s)
{
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(v[0]), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(v[1]), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v[0]), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(v[1]), "v"(s), "v"(h));
    const bool odd = (c & 2) != 0;                              // c % 4 == 2: this lane keeps the lo halves
    // the neighbour's words by DPP (quad_perm [1,0,3,2] = lane ^ 1; folds into the selects): no trip through the LDS crossbar
    const unsigned hn = (unsigned)__builtin_amdgcn_mov_dpp((int)h, 0xB1, 0xF, 0xF, true);
    const unsigned ln = (unsigned)__builtin_amdgcn_mov_dpp((int)l, 0xB1, 0xF, 0xF, true);
    return wino_u32x2{odd ? ln : h, odd ? l : hn};
}
// byte offset in the row: group of 8 channels = 32 bytes; hi halves of channels 0-3 | 4-7 at +0 | +8, lo at +16 | +24
__device__ __forceinline__ int split_pair_offset(int c) { return (c >> 3) * 32 + ((c >> 2) & 1) * 8 + ((c & 2) ? 16 : 0); }

__device__ __forceinline__ void store_split_pair(float *row, int c, f32x2 v, float s)
{
    wino_store(split_pair_words(c, v, s), reinterpret_cast<wino_u32x2 *>(reinterpret_cast<char *>(row) + split_pair_offset(c)));
}
// ... with the store's cache policy chosen by the caller (the pooler that also gathers from a map slice it wants to keep in L2)
template <bool NT>
__device__ __forceinline__ void store_split_pair_policy(float *row, int c, f32x2 v, float s)
{
    wino_u32x2 *p = reinterpret_cast<wino_u32x2 *>(reinterpret_cast<char *>(row) + split_pair_offset(c));
    const wino_u32x2 w = split_pair_words(c, v, s);
    if (NT)
        __builtin_nontemporal_store(w, p);
    else
        wino_store(w, p);
}

// GRAD = true applies A (x) A = (AT (x) AT)^T instead of BT (x) BT: the adjoint of the OUTPUT transform, which maps the
// gradient of a convolution's output into the transform domain (dM) for the weight gradient (locov_winograd_wgrad_f32).
template <bool GRAD>
__device__ __forceinline__ constexpr float in_coef(int f, int y) { return GRAD ? wino::AT[y][f] : wino::BT[f][y]; }

// Row fy of the transform of ONE 7x7 patch (two channels per lane): load(y, x) yields the patch value, emit(fx, v) takes the 11
// transform-domain values (fy, fx).  Only the patch rows with a non-zero coefficient for fy are touched (3 or 4 of the 7).
template <bool GRAD, int FY, typename Load, typename Emit>
__device__ __forceinline__ void wino_in_fy(Load &&load, Emit &&emit)
{
#pragma clang fp contract(fast)          // the same fused multiply-adds in every translation unit (roi_align_nhwc.hip is built with contraction off)
    f32x2 wv[7];
#pragma unroll
    for (int xx = 0; xx < 7; xx++) {
        f32x2 a = {0.f, 0.f};
#pragma unroll
        for (int y = 0; y < 7; y++)
            if (in_coef<GRAD>(FY, y) != 0.f) a += in_coef<GRAD>(FY, y) * load(y, xx);
        wv[xx] = a;
    }
#pragma unroll
    for (int fx = 0; fx < wino::NF; fx++) {
        f32x2 a = {0.f, 0.f};
#pragma unroll
        for (int xx = 0; xx < 7; xx++)
            if (in_coef<GRAD>(fx, xx) != 0.f) a += in_coef<GRAD>(fx, xx) * wv[xx];
        emit(fx, a);
    }
}

// fy = 0 .. 10 in order
template <bool GRAD, int FY = 0, typename Load, typename Emit>
__device__ __forceinline__ void wino_in_all(Load &&load, Emit &&emit)
{
    if constexpr (FY < wino::NF) {
        wino_in_fy<GRAD, FY>(load, [&](int fx, f32x2 v) { emit(FY, fx, v); });
        wino_in_all<GRAD, FY + 1>(load, emit);
    }
}

}  // namespace locov
