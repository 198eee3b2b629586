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


// ---- near-plane clipping (clip_near() in oracle/raster_oracle.c, same op order) ----
// A vertex is in front of the near plane iff d = z + w >= 0.  A triangle crossing the plane z = -w is cut into one or two pieces that
// are snapped / culled / covered / depth-tested like triangles of their own and carry the id (and, for the output values, the
// vertices) of the triangle they came from.  The cut point of an edge is always computed from the vertex in front towards the vertex
// behind, so that the two triangles sharing a mesh edge get the same point.
__device__ __forceinline__ float4 vhap_cut_edge(const float4 a, float da, const float4 b, float db) {
    const float t = __fdiv_rn(da, __fsub_rn(da, db));
    float4 c;
    c.x = __fmaf_rn(t, __fsub_rn(b.x, a.x), a.x);
    c.y = __fmaf_rn(t, __fsub_rn(b.y, a.y), a.y);
    c.w = __fmaf_rn(t, __fsub_rn(b.w, a.w), a.w);
    c.z = -c.w;                       // on the plane exactly: z/w == -1
    return c;
}

// bit i set: vertex i is behind the near plane
__device__ __forceinline__ int vhap_behind_mask(const float4 p0, const float4 p1, const float4 p2) {
    return (__fadd_rn(p0.z, p0.w) < 0.0f ? 1 : 0) | (__fadd_rn(p1.z, p1.w) < 0.0f ? 2 : 0) | (__fadd_rn(p2.z, p2.w) < 0.0f ? 4 : 0);
}

// Pieces of a triangle with 1 or 2 vertices behind the plane (behind = vhap_behind_mask, not 0 and not 7): returns 1 or 2;
// piece 0 = (a0, a1, a2), piece 1 = (b0, b1, b2).  No run-time indexed arrays (they would fall to scratch).
__device__ __forceinline__ int vhap_clip_near(const float4 p0, const float4 p1, const float4 p2, int behind, float4& a0, float4& a1,
                                              float4& a2, float4& b0, float4& b1, float4& b2) {
    const float d0 = __fadd_rn(p0.z, p0.w), d1 = __fadd_rn(p1.z, p1.w), d2 = __fadd_rn(p2.z, p2.w);
    if (__popc(behind) == 2) {        // one vertex in front: a, then b, c in winding order -> (a, ab, ac)
        const int k = behind == 6 ? 0 : (behind == 5 ? 1 : 2);
        const float4 a = k == 0 ? p0 : (k == 1 ? p1 : p2), b = k == 0 ? p1 : (k == 1 ? p2 : p0), c = k == 0 ? p2 : (k == 1 ? p0 : p1);
        const float da = k == 0 ? d0 : (k == 1 ? d1 : d2), db = k == 0 ? d1 : (k == 1 ? d2 : d0), dc = k == 0 ? d2 : (k == 1 ? d0 : d1);
        a0 = a;
        a1 = vhap_cut_edge(a, da, b, db);
        a2 = vhap_cut_edge(a, da, c, dc);
        b0 = b1 = b2 = a;
        return 1;
    }
    const int k = behind == 1 ? 0 : (behind == 2 ? 1 : 2);   // one vertex behind: o, then a, b in winding order -> (a, b, bo), (a, bo, ao)
    const float4 o = k == 0 ? p0 : (k == 1 ? p1 : p2), a = k == 0 ? p1 : (k == 1 ? p2 : p0), b = k == 0 ? p2 : (k == 1 ? p0 : p1);
    const float dO = k == 0 ? d0 : (k == 1 ? d1 : d2), da = k == 0 ? d1 : (k == 1 ? d2 : d0), db = k == 0 ? d2 : (k == 1 ? d0 : d1);
    const float4 bo = vhap_cut_edge(b, db, o, dO), ao = vhap_cut_edge(a, da, o, dO);
    a0 = a; a1 = b; a2 = bo;
    b0 = a; b1 = bo; b2 = ao;
    return 2;
}

// Conservative pixel bounding box of everything a triangle can cover, clipping included (for loops over a triangle's pixels that test
// the triangle id per pixel).  Returns false when nothing is drawn.
__device__ __forceinline__ bool tri_cover_bbox(const float4 p0, const float4 p1, const float4 p2, int H, int W, int& px0, int& px1,
                                               int& py0, int& py1) {
    int sx[3], sy[3];
    long long area;
    const int behind = vhap_behind_mask(p0, p1, p2);
    if (behind == 0) return tri_bbox(p0, p1, p2, H, W, sx, sy, area, px0, px1, py0, py1);
    if (behind == 7) return false;
    float4 a0, a1, a2, b0, b1, b2;
    const int np = vhap_clip_near(p0, p1, p2, behind, a0, a1, a2, b0, b1, b2);
    int qx0, qx1, qy0, qy1;
    const bool ok0 = tri_bbox(a0, a1, a2, H, W, sx, sy, area, px0, px1, py0, py1);
    const bool ok1 = np == 2 && tri_bbox(b0, b1, b2, H, W, sx, sy, area, qx0, qx1, qy0, qy1);
    if (ok0 && ok1) { px0 = min(px0, qx0); px1 = max(px1, qx1); py0 = min(py0, qy0); py1 = max(py1, qy1); }
    else if (ok1) { px0 = qx0; px1 = qx1; py0 = qy0; py1 = qy1; }
    return ok0 || ok1;
}
