// Area-weighted vertex normal of ONE vertex by gather over the static vertex->corner CSR (render_nvdiffrast.py compute_v_normals:
// n_raw[v] = sum over incident faces of (v1-v0) x (v2-v0); fallback (0,0,1) if |n|^2 <= 1e-20; normalise).  Shared by the stand-alone
// kernel (flame.hip) and by the extra workgroups of the binning launch (raster.hip: vhap_raster_bin_vnormal).
#pragma once
#include "common.h"

__device__ __forceinline__ void vhap_vnormal_vertex(const float* __restrict__ P, const int* __restrict__ tri, const int* __restrict__ vc_ptr,
                                                    const int* __restrict__ vc_idx, int v, float* __restrict__ o, float* __restrict__ inv_len) {
    float nx = 0.f, ny = 0.f, nz = 0.f;
    // the incident faces FOUR at a time (corner ids -> vertex ids -> positions: three dependent round trips per batch instead of per
    // face -- at valence ~6 this walk was ~18 of them in series and set the duration of the launch it rides in); summed in list order
    constexpr int FB = 4;
    static_assert(FB == 4, "the pinning statements below name four faces");
    const int k0 = vc_ptr[v], k1 = vc_ptr[v + 1];
    for (int k = k0; k < k1; k += FB) {
        int cc[FB], ii[FB][3];
        float p_[FB][3][3];
#pragma unroll
        for (int u = 0; u < FB; u++) cc[u] = vc_idx[k + u < k1 ? k + u : k1 - 1];
        // (the empty asm statements pin each batch: without them the compiler turns "face u exists" into a branch per face and sinks that
        // face's loads into it -- four chains of three round trips in series again, seen in the ISA of both kernels that use this)
        asm volatile("" ::"v"(cc[0]), "v"(cc[1]), "v"(cc[2]), "v"(cc[3]));
#pragma unroll
        for (int u = 0; u < FB; u++) {
            const int t = cc[u] / 3;
            ii[u][0] = tri[3 * t]; ii[u][1] = tri[3 * t + 1]; ii[u][2] = tri[3 * t + 2];
        }
#pragma unroll
        for (int u = 0; u < FB; u++) asm volatile("" ::"v"(ii[u][0]), "v"(ii[u][1]), "v"(ii[u][2]));
#pragma unroll
        for (int u = 0; u < FB; u++)
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int c = 0; c < 3; c++) p_[u][q][c] = P[3 * ii[u][q] + c];
#pragma unroll
        for (int u = 0; u < FB; u++)
#pragma unroll
            for (int q = 0; q < 3; q++) asm volatile("" ::"v"(p_[u][q][0]), "v"(p_[u][q][1]), "v"(p_[u][q][2]));
#pragma unroll
        for (int u = 0; u < FB; u++) {
            if (k + u >= k1) break;
            const float ax = p_[u][1][0] - p_[u][0][0], ay = p_[u][1][1] - p_[u][0][1], az = p_[u][1][2] - p_[u][0][2];
            const float bx = p_[u][2][0] - p_[u][0][0], by = p_[u][2][1] - p_[u][0][1], bz = p_[u][2][2] - p_[u][0][2];
            nx += ay * bz - az * by; ny += az * bx - ax * bz; nz += ax * by - ay * bx;
        }
    }
    float l2 = nx * nx + ny * ny + nz * nz;
    const bool fallback = !(l2 > 1e-20f);
    if (fallback) { nx = 0.f; ny = 0.f; nz = 1.f; l2 = 1.f; }
    const float inv = 1.0f / sqrtf(fmaxf(l2, 1e-20f));
    o[0] = nx * inv; o[1] = ny * inv; o[2] = nz * inv;
    if (inv_len) *inv_len = fallback ? 0.f : inv;    // saved for the backward: 1 / |raw normal| (0: constant fallback normal)
}
