// Device helpers shared by the split-operand GEMM kernels (gemm_split.hip: 128x128 tile, two workgroups per CU;
// gemm_split_big.hip: 256x256 tile, one 8-wave workgroup per CU).
#pragma once
#include "gemm_nt.h"

namespace locov {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 32;                      // fp32 columns per K-tile = the k of one v_mfma_f32_16x16x32_f16
constexpr int WROWB = 128;                  // split-layout tile rows in LDS: unpadded (LDS DMA writes 1 KB runs), XOR-swizzled

// W rows sit unpadded in LDS with their eight 16-byte chunks XOR-permuted by wswz(row) (a function of (row/2)%8):
// chosen so that the 16x16x32 fragment reads -- lane l: row l%16, k-group l/16 -- hit 16 distinct 16-byte slots of the
// 256-byte bank window in every 16-lane group the LDS serves at once ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32).
__device__ __forceinline__ int wswz(int row) { return (int)((0x75642031u >> (4 * ((row >> 1) & 7))) & 7u); }

__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// four fp32 values -> four hi halves and four lo halves of s*x, one vector-ALU instruction per half produced:
// v_fma_mix{lo,hi}_f16 forms a*b+c in fp32 from fp32 / f16 sources and rounds once to f16, so  hi = f16(x*s)  and
// lo = f16(x*s - hi)  (x*s and the difference are exact) take 8 instructions per chunk instead of the 14 the
// convert / multiply / subtract sequence compiles to -- the staging's vector-ALU work shares the SIMD's issue with the MFMAs.
__device__ __forceinline__ void split4(const f32x4 &x, float s, u32x2 &hi, u32x2 &lo)
{
#pragma unroll
    for (int e = 0; e < 2; e++) {
        unsigned h, l;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x[2 * e]), "s"(s));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x[2 * e + 1]), "s"(s));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x[2 * e]), "s"(s), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x[2 * e + 1]), "s"(s), "v"(h));
        hi[e] = h;
        lo[e] = l;
    }
}

// Epilogue traffic in the split layout (LOCOV_EPI_OUT_SPLIT / LOCOV_EPI_RES_SPLIT; scale = the launch's activation scale).  In
// the epilogue lane l owns the four columns c4 = 4 (l % 16) .. c4 + 3 of its row, i.e. the lane pair (l, l ^ 1) owns one group
// of 8 columns = 32 bytes of the layout: 8 hi halves (the even lane's 16 bytes), then 8 lo halves (the odd lane's).  So both
// directions are ONE 16-byte access per lane at the byte offset the fp32 value would have, plus one 8-byte exchange inside
// the pair.
__device__ __forceinline__ f32x4 unsplit4(const f32x4 &raw, bool odd, float inv_scale)
{
    const u32x4 w = __builtin_bit_cast(u32x4, raw);
    // even lane: hi of columns 0-3 = own words 0,1, lo = the odd lane's words 0,1; odd lane: hi of columns 4-7 = the even
    // lane's words 2,3, lo = own words 2,3
    const unsigned s0 = odd ? w[0] : w[2], s1 = odd ? w[1] : w[3];
    const unsigned r0 = (unsigned)__shfl_xor((int)s0, 1), r1 = (unsigned)__shfl_xor((int)s1, 1);
    const unsigned h0 = odd ? r0 : w[0], h1 = odd ? r1 : w[1], l0 = odd ? w[2] : r0, l1 = odd ? w[3] : r1;
    const f16x4 hv = __builtin_bit_cast(f16x4, u32x2{h0, h1}), lv = __builtin_bit_cast(f16x4, u32x2{l0, l1});
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = ((float)hv[j] + (float)lv[j]) * inv_scale;      // hi + lo is exact in fp32 (<= 23 bits apart)
    return o;
}

__device__ __forceinline__ u32x4 split4_pair(const f32x4 &v, bool odd, float scale)
{
    u32x2 hi, lo;
    split4(v, scale, hi, lo);
    // the even lane stores the 8 hi halves of the group (its own 4, then the odd lane's), the odd lane the 8 lo halves
    const unsigned s0 = odd ? hi[0] : lo[0], s1 = odd ? hi[1] : lo[1];
    const unsigned r0 = (unsigned)__shfl_xor((int)s0, 1), r1 = (unsigned)__shfl_xor((int)s1, 1);
    return odd ? u32x4{r0, r1, lo[0], lo[1]} : u32x4{hi[0], hi[1], r0, r1};
}

}  // namespace

}  // namespace locov
