// Sampling arithmetic shared by the channels-last ROIAlign kernels (roi_align_nhwc.hip, roi_align_tiles.hip): torchvision's
// roi_align as reached from ovr/modeling/roi_heads/roi_emb_heads.py:243-245, every step an explicitly rounded fp32 operation
// (these sources are built with -ffp-contract=off: the coordinates must be the CPU oracle's, bit for bit).
#pragma once
#include "common.h"

namespace locov {

struct AxisSampleN {
    int lo, hi;   // pixel index along the axis
    float wl, wh;
};

// sample i of bin p on one axis: coordinate, the [-1, size] rule (weights 0), the clamp at 0 and at the last pixel
__device__ __forceinline__ AxisSampleN axis_sample_n(float start, float bin, int p, int i, int grid, int size)
{
    float v = __fadd_rn(__fadd_rn(start, __fmul_rn((float)p, bin)),
                        __fdiv_rn(__fmul_rn(__fadd_rn((float)i, .5f), bin), (float)grid));
    AxisSampleN s;
    if (v < -1.0f || v > (float)size) {
        s.lo = 0; s.hi = 0; s.wl = 0.f; s.wh = 0.f;
        return s;
    }
    if (v <= 0.f) v = 0.f;
    int lo = (int)v, hi;
    if (lo >= size - 1) {
        hi = lo = size - 1;
        v = (float)lo;
    } else {
        hi = lo + 1;
    }
    const float l = __fsub_rn(v, (float)lo);
    s.lo = lo; s.hi = hi; s.wl = l; s.wh = __fsub_rn(1.f, l);
    return s;
}

}  // namespace locov
