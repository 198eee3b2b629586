// Probe for tests/test_host_cpu.py::test_sh_diffuse_is_not_contracted: k_full shades, k_norm only normalises (the correctly-rounded
// sqrt / divide expansions contain fused multiply-adds of their own); any fused instruction k_full has beyond k_norm's is a contraction
// inside sh_diffuse.
#include "shade_common.h"
extern "C" __global__ void k_full(const float* n, const float* sc, const float* l, float* out) {
    SH9 b;
    float x, y, z, inv, d[3];
    sh_diffuse(n[threadIdx.x * 3], n[threadIdx.x * 3 + 1], n[threadIdx.x * 3 + 2], sc, l, b, x, y, z, inv, d);
    out[threadIdx.x * 3] = d[0]; out[threadIdx.x * 3 + 1] = d[1]; out[threadIdx.x * 3 + 2] = d[2];
}
#pragma clang fp contract(off)
extern "C" __global__ void k_norm(const float* n, float* out) {
    const float nx = n[threadIdx.x * 3], ny = n[threadIdx.x * 3 + 1], nz = n[threadIdx.x * 3 + 2];
    out[threadIdx.x] = 1.0f / __builtin_sqrtf(fmaxf(((nx * nx) + (ny * ny)) + (nz * nz), 1e-20f));
}
