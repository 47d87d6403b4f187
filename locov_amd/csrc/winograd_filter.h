// The 1-D filter transform of the Winograd-domain 3x3 convolution, (G g)[f] = G[f][0] g0 + G[f][1] g1 + G[f][2] g2 in fp64, as
// ONE explicit operation sequence (a product and two fused multiply-adds): the filter transform is evaluated by two kernels --
// wino_pack_weight_kernel (winograd.hip) and the one-launch operand preparation of a training step (weight_prep.hip) -- whose
// results must be the same bits, which the compiler's own choice of contractions does not promise.
#pragma once
#include "winograd_tables.h"

namespace locov {
namespace wino {

// Four of the eleven transform points (0 and infinity of both sub-transforms) have ONE non-zero coefficient: their "dot product" is a
// single multiplication (the skipped terms are +-0 for finite data) -- a third of the one-launch preparation's fp64 work, which is what
// bounds it (weight_prep.hip: ~460 fp64 operations per filter and operand).
__device__ __forceinline__ double filter_dot3(int f, double g0, double g1, double g2)
{
    if (G[f][1] == 0.0 && G[f][2] == 0.0) return G[f][0] * g0;
    if (G[f][0] == 0.0 && G[f][1] == 0.0) return G[f][2] * g2;
    return fma(G[f][2], g2, fma(G[f][1], g1, G[f][0] * g0));
}

}  // namespace wino
}  // namespace locov
