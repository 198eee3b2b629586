// Per-fragment arithmetic of the G-buffer pass, shared by the rasteriser (raster.hip: plain / fused-interpolate / deferred-shading
// modes) and the deferred backward (deferred.hip), which RE-COMPUTES the interpolated normal / uv / uv derivatives of a pixel from its
// (triangle id, geometry) instead of reading them back: both must produce the same bits, so the code lives here once.  Same op order
// as shade_frag() / interpolate in oracle/raster_oracle.c; every fused multiply-add is an explicit intrinsic and contraction is off.
#pragma once
#include "common.h"

namespace {

struct Frag {
    float b0, b1, zw, iw;
};

#pragma clang fp contract(off)
__device__ __forceinline__ Frag shade_frag(const float4 p0, const float4 p1, const float4 p2, float fx, float fy) {
    Frag r;
    const float p0x = __fmaf_rn(-fx, p0.w, p0.x), p0y = __fmaf_rn(-fy, p0.w, p0.y);
    const float p1x = __fmaf_rn(-fx, p1.w, p1.x), p1y = __fmaf_rn(-fy, p1.w, p1.y);
    const float p2x = __fmaf_rn(-fx, p2.w, p2.x), p2y = __fmaf_rn(-fy, p2.w, p2.y);
    const float a0 = __fmaf_rn(p1x, p2y, -(p1y * p2x));
    const float a1 = __fmaf_rn(p2x, p0y, -(p2y * p0x));
    const float a2 = __fmaf_rn(p0x, p1y, -(p0y * p1x));
    const float at = (a0 + a1) + a2;
    const float iw = (fabsf(at) > 0.0f) ? __fdiv_rn(1.0f, at) : 0.0f;
    const float z = __fmaf_rn(p0.z, a0, __fmaf_rn(p1.z, a1, p2.z * a2));
    const float w = __fmaf_rn(p0.w, a0, __fmaf_rn(p1.w, a1, p2.w * a2));
    const float zw = __fdiv_rn(z, w);
    r.b0 = fminf(fmaxf(a0 * iw, 0.0f), 1.0f);
    r.b1 = fminf(fmaxf(a1 * iw, 0.0f), 1.0f);
    r.zw = fminf(fmaxf(zw, -1.0f), 1.0f);
    r.iw = iw;
    return r;
}

// screen-space derivatives of the barycentrics: (du/dX, du/dY, dv/dX, dv/dY) in pixel units
__device__ __forceinline__ float4 frag_db(const float4 p0, const float4 p1, const float4 p2, const Frag fr, float xs, float ys) {
    const float dfxdx = xs * fr.iw, dfydy = ys * fr.iw;
    const float da0dx = __fmaf_rn(p2.y, p1.w, -(p1.y * p2.w));
    const float da0dy = __fmaf_rn(p1.x, p2.w, -(p2.x * p1.w));
    const float da1dx = __fmaf_rn(p0.y, p2.w, -(p2.y * p0.w));
    const float da1dy = __fmaf_rn(p2.x, p0.w, -(p0.x * p2.w));
    const float da2dx = __fmaf_rn(p1.y, p0.w, -(p0.y * p1.w));
    const float da2dy = __fmaf_rn(p0.x, p1.w, -(p1.x * p0.w));
    const float datdx = (da0dx + da1dx) + da2dx;
    const float datdy = (da0dy + da1dy) + da2dy;
    float4 o_db;
    o_db.x = dfxdx * __fmaf_rn(fr.b0, datdx, -da0dx);
    o_db.y = dfydy * __fmaf_rn(fr.b0, datdy, -da0dy);
    o_db.z = dfxdx * __fmaf_rn(fr.b1, datdx, -da1dx);
    o_db.w = dfydy * __fmaf_rn(fr.b1, datdy, -da1dy);
    return o_db;
}

struct FragAttr {
    float n0, n1, n2;   // interpolated (un-normalised) vertex normal
    float tu, tv;       // texture coordinate
    float4 td;          // (du/dX, du/dY, dv/dX, dv/dY) of the texture coordinate
};

// dr.interpolate(v_normal) and dr.interpolate(uv, diff_attrs='all') for one fragment.  N: this frame's vertex normals [V,3];
// UV [VT,2]; (i0,i1,i2) vertex ids, (j0,j1,j2) uv-vertex ids of the triangle
__device__ __forceinline__ FragAttr frag_attr(const float* __restrict__ N, const float2* __restrict__ UV, int i0, int i1, int i2, int j0,
                                              int j1, int j2, const Frag fr, const float4 o_db) {
    FragAttr a;
    const float b2 = (1.0f - fr.b0) - fr.b1;
    a.n0 = __fmaf_rn(fr.b0, N[3 * i0 + 0], __fmaf_rn(fr.b1, N[3 * i1 + 0], b2 * N[3 * i2 + 0]));
    a.n1 = __fmaf_rn(fr.b0, N[3 * i0 + 1], __fmaf_rn(fr.b1, N[3 * i1 + 1], b2 * N[3 * i2 + 1]));
    a.n2 = __fmaf_rn(fr.b0, N[3 * i0 + 2], __fmaf_rn(fr.b1, N[3 * i1 + 2], b2 * N[3 * i2 + 2]));
    const float2 u0 = UV[j0], u1 = UV[j1], u2 = UV[j2];
    a.tu = __fmaf_rn(fr.b0, u0.x, __fmaf_rn(fr.b1, u1.x, b2 * u2.x));
    a.tv = __fmaf_rn(fr.b0, u0.y, __fmaf_rn(fr.b1, u1.y, b2 * u2.y));
    const float eu0 = u0.x - u2.x, eu1 = u1.x - u2.x, ev0 = u0.y - u2.y, ev1 = u1.y - u2.y;
    a.td.x = __fmaf_rn(o_db.x, eu0, o_db.z * eu1);
    a.td.y = __fmaf_rn(o_db.y, eu0, o_db.w * eu1);
    a.td.z = __fmaf_rn(o_db.x, ev0, o_db.z * ev1);
    a.td.w = __fmaf_rn(o_db.y, ev0, o_db.w * ev1);
    return a;
}
#pragma clang fp contract(fast)

}  // namespace
