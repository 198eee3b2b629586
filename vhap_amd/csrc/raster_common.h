// Snapping / culling / bounding-box rules shared by the rasteriser and the triangle-parallel backward
// (conventions of DESIGN.md section 3; every float op is an explicit intrinsic, so the result does not depend on
// the including file's contraction mode).
#pragma once
#include "common.h"

constexpr float VHAP_GUARD = 1048576.0f;  // 2^20 sub-pixel units

__device__ __forceinline__ bool snap_vertex(const float4 p, float hw, float hh, int& sx, int& sy) {
    if (!(p.w > 0.0f)) return false;
    const float xn = __fdiv_rn(p.x, p.w), yn = __fdiv_rn(p.y, p.w);
    const float fx = __fmaf_rn(xn, hw, hw), fy = __fmaf_rn(yn, hh, hh);
    if (!(fabsf(fx) < VHAP_GUARD) || !(fabsf(fy) < VHAP_GUARD)) return false;
    sx = __float2int_rn(fx);
    sy = __float2int_rn(fy);
    return true;
}

// Snap + cull + pixel bbox (inclusive, clipped to the image).  Returns false when nothing to draw.
__device__ __forceinline__ bool tri_bbox(const float4 p0, const float4 p1, const float4 p2, int H, int W,
                                         int (&sx)[3], int (&sy)[3], long long& area, int& px0, int& px1, int& py0,
                                         int& py1) {
    const float hw = 8.0f * (float)W, hh = 8.0f * (float)H;
    if (!snap_vertex(p0, hw, hh, sx[0], sy[0])) return false;
    if (!snap_vertex(p1, hw, hh, sx[1], sy[1])) return false;
    if (!snap_vertex(p2, hw, hh, sx[2], sy[2])) return false;
    area = (long long)(sx[1] - sx[0]) * (sy[2] - sy[0]) - (long long)(sx[2] - sx[0]) * (sy[1] - sy[0]);
    if (area <= 0) return false;  // back-facing or degenerate
    const int minx = min(sx[0], min(sx[1], sx[2])), maxx = max(sx[0], max(sx[1], sx[2]));
    const int miny = min(sy[0], min(sy[1], sy[2])), maxy = max(sy[0], max(sy[1], sy[2]));
    px0 = max((minx - 8 + 15) >> 4, 0);
    px1 = min((maxx - 8) >> 4, W - 1);
    py0 = max((miny - 8 + 15) >> 4, 0);
    py1 = min((maxy - 8) >> 4, H - 1);
    return px0 <= px1 && py0 <= py1;
}

