// Layout of the in-place antialiasing's pair list (antialias.hip writes it, the photometric sum's launch reads it: pixel.hip) and the
// colour part of the antialias backward for the photometric loss, shared by both.
#pragma once
#include "common.h"

namespace {

// work[0] = number of items; items from work + 4, AA_ITEM2 ints each: int4 (pixel index pi, direction | flags, alpha bits, frame),
// float4 c0, float4 c1 (the two colours the blend saw)
constexpr int AA_ITEM2 = 12;

// d L / d out[q] = -sign(gt - out[q]) on rgb per unit of the upstream photometric gradient (tracker.py:430-439); the colour part of the
// antialias backward -- what flows to the pair's two pixels BESIDES the pass-through -- is +-alpha times it.  UNSCALED: the consumer
// (deferred_shade_bwd with VHAP_CALL_DELTA_UNSCALED) multiplies by the upstream gradient d_sum, which is known only when the photometric
// sum has finished -- so this part can run in the SAME launch as the sum instead of behind it.
__device__ __forceinline__ void aa_colour_bwd_item(const int* __restrict__ work, int i, const float4* __restrict__ pred,
                                                   const float* __restrict__ gt, int H, int W, float* __restrict__ d_delta) {
    const int4 h = (reinterpret_cast<const int4*>(work + 4) + (size_t)i * (AA_ITEM2 / 4))[0];
    const int HW = H * W;
    const long long pi = (unsigned)h.x;
    const int d = h.y & 1;
    const float alpha = __int_as_float(h.z);
    const int b = h.w;
    const long long pj = pi + (d == 0 ? 1 : W);
    const long long q = alpha > 0.0f ? pi : pj;
    const int remq = (int)(q - (long long)b * HW);
    const int qy = remq / W, qx = remq - qy * W;
    const float* g = gt + (size_t)b * 3 * HW + (size_t)(H - 1 - qy) * W + qx;
    const float4 p = pred[q];
    auto sg = [](float e) { return e > 0.f ? 1.0f : (e < 0.f ? -1.0f : 0.0f); };
    const float go[3] = {-sg(g[0] - p.x), -sg(g[HW] - p.y), -sg(g[2 * HW] - p.z)};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        atomicAdd(&d_delta[(size_t)pj * 4 + k], alpha * go[k]);
        atomicAdd(&d_delta[(size_t)pi * 4 + k], -alpha * go[k]);
    }
}

}  // namespace
