// Area-weighted vertex normal of ONE vertex by gather over the static vertex->corner CSR (render_nvdiffrast.py compute_v_normals:
// n_raw[v] = sum over incident faces of (v1-v0) x (v2-v0); fallback (0,0,1) if |n|^2 <= 1e-20; normalise).  Shared by the stand-alone
// kernel (flame.hip) and by the extra workgroups of the binning launch (raster.hip: vhap_raster_bin_vnormal).
#pragma once
#include "common.h"

__device__ __forceinline__ void vhap_vnormal_vertex(const float* __restrict__ P, const int* __restrict__ tri, const int* __restrict__ vc_ptr,
                                                    const int* __restrict__ vc_idx, int v, float* __restrict__ o, float* __restrict__ inv_len) {
    float nx = 0.f, ny = 0.f, nz = 0.f;
    for (int k = vc_ptr[v]; k < vc_ptr[v + 1]; k++) {
        const int t = vc_idx[k] / 3;
        const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
        const float ax = P[3 * i1] - P[3 * i0], ay = P[3 * i1 + 1] - P[3 * i0 + 1], az = P[3 * i1 + 2] - P[3 * i0 + 2];
        const float bx = P[3 * i2] - P[3 * i0], by = P[3 * i2 + 1] - P[3 * i0 + 1], bz = P[3 * i2 + 2] - P[3 * i0 + 2];
        nx += ay * bz - az * by; ny += az * bx - ax * bz; nz += ax * by - ay * bx;
    }
    float l2 = nx * nx + ny * ny + nz * nz;
    const bool fallback = !(l2 > 1e-20f);
    if (fallback) { nx = 0.f; ny = 0.f; nz = 1.f; l2 = 1.f; }
    const float inv = 1.0f / sqrtf(fmaxf(l2, 1e-20f));
    o[0] = nx * inv; o[1] = ny * inv; o[2] = nz * inv;
    if (inv_len) *inv_len = fallback ? 0.f : inv;    // saved for the backward: 1 / |raw normal| (0: constant fallback normal)
}
