// The 1-D filter transform of the Winograd-domain 3x3 convolution, (G g)[f] = G[f][0] g0 + G[f][1] g1 + G[f][2] g2 in fp64, as
// ONE explicit operation sequence (a product and two fused multiply-adds): the filter transform is evaluated by two kernels --
// wino_pack_weight_kernel (winograd.hip) and the one-launch operand preparation of a training step (weight_prep.hip) -- whose
// results must be the same bits, which the compiler's own choice of contractions does not promise.
#pragma once
#include "winograd_tables.h"

namespace locov {
namespace wino {

__device__ __forceinline__ double filter_dot3(int f, double g0, double g1, double g2)
{
    return fma(G[f][2], g2, fma(G[f][1], g1, G[f][0] * g0));
}

}  // namespace wino
}  // namespace locov
