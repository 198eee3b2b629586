// 9-band spherical-harmonics diffuse shading of one pixel (render_nvdiffrast.py:19-53 get_SH_shading, :386 safe_normalize), shared
// by the stand-alone shading kernels (pixel.hip) and the deferred-shading kernels (raster.hip forward, deferred.hip backward).
#pragma once
#include "common.h"

namespace {

struct SH9 {
    float v[9];
};

// Contraction is OFF for sh_basis / sh_diffuse (and every operation is a plain operator INSIDE the pragma region -- the __fmul_rn / __fadd_rn
// "intrinsics" of the HIP headers are ordinary inline functions compiled under the header's own contraction mode and fuse again after
// inlining): the diffuse value of a pixel is computed by the forward pass (raster.hip mode 2 / pixel.hip), which records its maximum for
// the diffuse regulariser, and RE-computed by the backward (deferred.hip, pixel.hip), which routes the regulariser's max-gradient to
// the pixels whose value EQUALS the recorded maximum (tracker.py:547-550: relu(diffuse.max() - 1)).  Left to the compiler, the
// contraction decisions of two kernels differ, the re-computed maximum misses the recorded one by an ulp in a fifth of the evaluations,
// and the max term silently vanishes from d(lights) -- round 2's "flaky" fit parity (profiles/r03_fit_flake_hunt_*.txt).
// tests/test_host_cpu.py::test_sh_diffuse_is_not_contracted disassembles a probe kernel and counts the fused instructions.
#pragma clang fp contract(off)
__device__ __forceinline__ void sh_basis(float x, float y, float z, const float* __restrict__ sc, SH9& b) {
    b.v[0] = sc[0];
    b.v[1] = x * sc[1];
    b.v[2] = y * sc[2];
    b.v[3] = z * sc[3];
    b.v[4] = (x * y) * sc[4];
    b.v[5] = (x * z) * sc[5];
    b.v[6] = (y * z) * sc[6];
    b.v[7] = ((x * x) - (y * y)) * sc[7];
    b.v[8] = (((3.0f * z) * z) - 1.0f) * sc[8];
}

// raw normal -> normalised direction (x, y, z), 1 / max(|r|, 1e-10), diffuse colour d[3]; l [9,3] lights, sc [9] constants
__device__ __forceinline__ void sh_diffuse(float nx, float ny, float nz, const float* __restrict__ sc, const float* __restrict__ l, SH9& b,
                                           float& x, float& y, float& z, float& inv, float (&d)[3]) {
    const float l2 = ((nx * nx) + (ny * ny)) + (nz * nz);
    inv = 1.0f / __builtin_sqrtf(fmaxf(l2, 1e-20f));
    x = nx * inv; y = ny * inv; z = nz * inv;
    sh_basis(x, y, z, sc, b);
    d[0] = d[1] = d[2] = 0.f;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const float t0 = b.v[k] * l[3 * k], t1 = b.v[k] * l[3 * k + 1], t2 = b.v[k] * l[3 * k + 2];
        d[0] = d[0] + t0;
        d[1] = d[1] + t1;
        d[2] = d[2] + t2;
    }
}
#pragma clang fp contract(fast)

// order-preserving float -> unsigned (negative ? ~u : u | sign)
__device__ __forceinline__ unsigned sh_f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// (max << 32 | tie count) monoid of the diffuse-regulariser statistic (tracker.py:547-550: diffuse.max() with torch's even split of the
// gradient among ties)
__device__ __forceinline__ unsigned long long sh_merge_max(unsigned long long a, unsigned long long b) {
    const unsigned ha = (unsigned)(a >> 32), hb = (unsigned)(b >> 32);
    return ha > hb ? a : (hb > ha ? b : a + (b & 0xffffffffull));
}

// gradient of the diffuse colour w.r.t. the RAW normal: gd[3] = d L / d diffuse -> (gnx, gny, gnz)
__device__ __forceinline__ void sh_normal_bwd(float x, float y, float z, float inv, bool clampd, const float* __restrict__ sc,
                                              const float* __restrict__ l, const float (&gd)[3], float& gnx, float& gny, float& gnz) {
    float gb[9];   // d L / d(basis_k / const_k)
#pragma unroll
    for (int k = 0; k < 9; k++) gb[k] = sc[k] * (l[3 * k] * gd[0] + l[3 * k + 1] * gd[1] + l[3 * k + 2] * gd[2]);
    gnx = gb[1] + y * gb[4] + z * gb[5] + 2.f * x * gb[7];
    gny = gb[2] + x * gb[4] + z * gb[6] - 2.f * y * gb[7];
    gnz = gb[3] + x * gb[5] + y * gb[6] + 6.f * z * gb[8];
    // n = r / max(|r|, 1e-10):  d r = (d n - n (n . d n)) / |r|   (or d n / 1e-10 when clamped)
    if (!clampd) {
        const float dot = x * gnx + y * gny + z * gnz;
        gnx = (gnx - x * dot) * inv; gny = (gny - y * dot) * inv; gnz = (gnz - z * dot) * inv;
    } else {
        gnx *= inv; gny *= inv; gnz *= inv;
    }
}

}  // namespace
