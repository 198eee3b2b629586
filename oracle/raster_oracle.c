/*
 * oracle/raster_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, never shipped, never the
 * product path) of the rasterize + interpolate semantics of the hot path.
 *
 * What it restates
 * ----------------
 * The reference (ShenhanQian/VHAP) calls `dr.rasterize` / `dr.interpolate` from the third-party
 * CUDA library nvdiffrast (fork ShenhanQian/nvdiffrast, branch `backface-culling`, pinned by
 * branch name only in /root/reference/pyproject.toml:30; NOT vendored, NOT installed) at
 *   vhap/util/render_nvdiffrast.py:254   dr.rasterize(ctx, pos[B,V,4], tri[F,3], (H,W))
 *   vhap/util/render_nvdiffrast.py:384   dr.interpolate(attr, rast, tri)
 *   vhap/util/render_nvdiffrast.py:389   dr.interpolate(uv[None], rast, tri_uv, rast_db, 'all')
 * The per-pixel arithmetic below (perspective-correct barycentrics from clip-space edge
 * functions, z/w, the analytic pixel differentials, `u*a0 + v*a1 + (1-u-v)*a2`) follows the
 * published algorithm (Laine et al., "Modular Primitives for High-Performance Differentiable
 * Rendering", SIGGRAPH Asia 2020, sections 3.2-3.3 and the nvdiffrast documentation).
 *
 * PARITY UNPINNED: the reference has no tests / golden vectors for this path and nvdiffrast
 * cannot run in this environment, so the *visibility conventions* are specified here and are
 * the contract the HIP kernels are checked against bit-for-bit:
 *   - output row 0 is the BOTTOM row (y-up NDC), pixel (px,py) centre at NDC
 *       fx = fma(2/W, px, 1/W - 1),  fy = fma(2/H, py, 1/H - 1)
 *   - vertices are snapped to a 1/16-pixel grid: sx = rint(fma(x/w, 8W, 8W)) (ties-to-even),
 *     pixel centre px sits at 16*px+8; a piece is dropped when any w <= 0 or any snapped
 *     coordinate magnitude >= 2^20 (guard band)
 *   - NEAR-PLANE CLIPPING (nvdiffrast's rasteriser clips in homogeneous clip space): a vertex is
 *     in front of the near plane iff d = z + w >= 0 (one float add).  A triangle with no vertex
 *     behind is drawn as is; with all three behind it is dropped; otherwise it is cut by the plane
 *     z = -w into ONE or TWO pieces (clip_near() below) that are snapped, culled, covered and
 *     depth-tested like triangles of their own but carry the id of the triangle they came from.
 *     The cut point on an edge is always computed from the vertex in front (a) towards the vertex
 *     behind (b): t = da / (da - db), c = fma(t, b - a, a) for x, y, w and z = -w -- so the two
 *     triangles sharing a mesh edge get the SAME cut point and the mesh stays watertight.  One
 *     vertex in front (a; b, c follow it in the triangle's winding): piece (a, ab, ac).  Two in
 *     front (a, b; o behind, winding o -> a -> b): pieces (a, b, bo) and (a, bo, ao).
 *     The OUTPUT barycentrics / z/w / differentials of a winning pixel are those of the ORIGINAL
 *     triangle (the homogeneous formulas of shade_frag() hold on both sides of w = 0)
 *   - coverage uses exact integer edge functions on the snapped coordinates; a pixel on an edge
 *     (E == 0) is inside iff A > 0 || (A == 0 && B > 0) for E = A*x + B*y + C
 *   - back-face culling: snapped signed area <= 0 is culled (CCW = front in y-up NDC)
 *   - depth TEST value: z/w is affine in screen space, so it is evaluated from the plane through
 *     the three snapped vertices carrying zw_i = z_i / w_i, anchored at the origin of the
 *     aligned 8x8-pixel block that contains the pixel (depth_plane() below: double-precision
 *     setup from the exact integer edge functions, rounded to float, then two float FMAs per
 *     pixel); fragments whose test value is outside [-1,1] (or NaN) are discarded; the nearest
 *     value wins, exact ties go to the LOWEST triangle index
 *   - the z/w that is OUTPUT (rast[...,2]) is recomputed for the winner with the perspective
 *     formula below (shade_frag) and clamped to [-1,1]
 *   - output rast = (u, v, z/w, float(tri+1)), u weights vertex 0, v vertex 1; empty = 0
 * All float arithmetic is written with explicit fmaf() and compiled with -ffp-contract=off so
 * that it is bit-reproducible on any IEEE-754 machine (the HIP kernel uses the same op order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GUARD 1048576.0f /* 2^20 sub-pixel units */

typedef struct { int x[3], y[3]; int ok; } snapped_t;

static inline uint32_t f2ord(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

static int snap_tri(const float* p0, const float* p1, const float* p2, int H, int W, snapped_t* s) {
    const float* p[3] = {p0, p1, p2};
    const float hw = 8.0f * (float)W, hh = 8.0f * (float)H;
    for (int i = 0; i < 3; i++) {
        float w = p[i][3];
        if (!(w > 0.0f)) return 0;
        float xn = p[i][0] / w, yn = p[i][1] / w;
        float sx = fmaf(xn, hw, hw), sy = fmaf(yn, hh, hh);
        if (!(fabsf(sx) < GUARD) || !(fabsf(sy) < GUARD)) return 0;
        s->x[i] = (int)lrintf(sx);
        s->y[i] = (int)lrintf(sy);
    }
    return 1;
}

/* winner arithmetic (output values) */
typedef struct { float b0, b1, zw, iw; } frag_t;

static inline frag_t shade_frag(const float* p0, const float* p1, const float* p2, float fx, float fy) {
    frag_t r;
    float p0x = fmaf(-fx, p0[3], p0[0]), p0y = fmaf(-fy, p0[3], p0[1]);
    float p1x = fmaf(-fx, p1[3], p1[0]), p1y = fmaf(-fy, p1[3], p1[1]);
    float p2x = fmaf(-fx, p2[3], p2[0]), p2y = fmaf(-fy, p2[3], p2[1]);
    float a0 = fmaf(p1x, p2y, -(p1y * p2x));
    float a1 = fmaf(p2x, p0y, -(p2y * p0x));
    float a2 = fmaf(p0x, p1y, -(p0y * p1x));
    float at = (a0 + a1) + a2;
    float iw = (fabsf(at) > 0.0f) ? 1.0f / at : 0.0f;
    float z = fmaf(p0[2], a0, fmaf(p1[2], a1, p2[2] * a2));
    float w = fmaf(p0[3], a0, fmaf(p1[3], a1, p2[3] * a2));
    float zw = z / w;
    r.b0 = fminf(fmaxf(a0 * iw, 0.0f), 1.0f);
    r.b1 = fminf(fmaxf(a1 * iw, 0.0f), 1.0f);
    r.zw = fminf(fmaxf(zw, -1.0f), 1.0f);
    r.iw = iw;
    return r;
}

/* depth-test plane of one triangle anchored at block origin (bx0,by0) (multiples of 8 pixels).
 * A1,B1 / A2,B2: coefficients of the edge functions opposite vertex 1 / 2 (weights of v1 / v2). */
typedef struct { float zwc, gx, gy; } zplane_t;

static inline zplane_t depth_plane(const float* p0, const float* p1, const float* p2, const snapped_t* s,
                                   int64_t area, int bx0, int by0) {
    zplane_t r;
    float zw0 = p0[2] / p0[3], zw1 = p1[2] / p1[3], zw2 = p2[2] / p2[3];
    int64_t cx = 16 * (int64_t)bx0 + 8, cy = 16 * (int64_t)by0 + 8;
    int64_t A1 = (int64_t)s->y[2] - s->y[0], B1 = (int64_t)s->x[0] - s->x[2];
    int64_t A2 = (int64_t)s->y[0] - s->y[1], B2 = (int64_t)s->x[1] - s->x[0];
    int64_t E1 = A1 * (cx - s->x[2]) + B1 * (cy - s->y[2]);
    int64_t E2 = A2 * (cx - s->x[0]) + B2 * (cy - s->y[0]);
    double d1 = (double)zw1 - (double)zw0, d2 = (double)zw2 - (double)zw0;
    double inv = 1.0 / (double)area;
    r.zwc = (float)((double)zw0 + ((double)E1 * d1 + (double)E2 * d2) * inv);
    r.gx = (float)((((double)A1 * d1 + (double)A2 * d2) * 16.0) * inv);
    r.gy = (float)((((double)B1 * d1 + (double)B2 * d2) * 16.0) * inv);
    return r;
}

/* Near-plane clipping of one triangle (see the header).  q receives 0, 1 or 2 pieces of three
 * xyzw vertices each; returns the number of pieces.  A triangle entirely in front is returned
 * unchanged (same bits as without clipping). */
static inline void cut_edge(const float* a, float da, const float* b, float db, float* c) {
    float t = da / (da - db);          /* a in front (da >= 0), b behind (db < 0): 0 <= t < 1 */
    c[0] = fmaf(t, b[0] - a[0], a[0]);
    c[1] = fmaf(t, b[1] - a[1], a[1]);
    c[3] = fmaf(t, b[3] - a[3], a[3]);
    c[2] = -c[3];                      /* on the plane z = -w exactly: z/w == -1 */
}

static int clip_near(const float* p0, const float* p1, const float* p2, float q[2][3][4]) {
    const float* p[3] = {p0, p1, p2};
    float d[3];
    int behind = 0, nb = 0;
    for (int i = 0; i < 3; i++) {
        d[i] = p[i][2] + p[i][3];
        if (d[i] < 0.0f) { behind |= 1 << i; nb++; }
    }
    if (nb == 0) {
        for (int i = 0; i < 3; i++) memcpy(q[0][i], p[i], 16);
        return 1;
    }
    if (nb == 3) return 0;
    if (nb == 2) {                     /* one vertex in front: a, then b, c in winding order */
        int k = (behind == 6) ? 0 : (behind == 5) ? 1 : 2;
        const float *a = p[k], *b = p[(k + 1) % 3], *c = p[(k + 2) % 3];
        memcpy(q[0][0], a, 16);
        cut_edge(a, d[k], b, d[(k + 1) % 3], q[0][1]);
        cut_edge(a, d[k], c, d[(k + 2) % 3], q[0][2]);
        return 1;
    }
    {                                  /* one vertex behind: o, then a, b in winding order */
        int k = (behind == 1) ? 0 : (behind == 2) ? 1 : 2;
        const float *o = p[k], *a = p[(k + 1) % 3], *b = p[(k + 2) % 3];
        float bo[4], ao[4];
        cut_edge(b, d[(k + 2) % 3], o, d[k], bo);
        cut_edge(a, d[(k + 1) % 3], o, d[k], ao);
        memcpy(q[0][0], a, 16); memcpy(q[0][1], b, 16); memcpy(q[0][2], bo, 16);
        memcpy(q[1][0], a, 16); memcpy(q[1][1], bo, 16); memcpy(q[1][2], ao, 16);
        return 2;
    }
}

/* coverage + depth test of one piece (a whole triangle or a piece of a clipped one) carrying triangle id t */
static void raster_piece(const float* p0, const float* p1, const float* p2, int t, int H, int W, uint64_t* vis) {
    snapped_t s;
    if (!snap_tri(p0, p1, p2, H, W, &s)) return;
    int64_t area = (int64_t)(s.x[1] - s.x[0]) * (s.y[2] - s.y[0]) - (int64_t)(s.x[2] - s.x[0]) * (s.y[1] - s.y[0]);
    if (area <= 0) return; /* back-facing or degenerate */
    int minx = s.x[0], maxx = s.x[0], miny = s.y[0], maxy = s.y[0];
    for (int i = 1; i < 3; i++) {
        if (s.x[i] < minx) minx = s.x[i]; if (s.x[i] > maxx) maxx = s.x[i];
        if (s.y[i] < miny) miny = s.y[i]; if (s.y[i] > maxy) maxy = s.y[i];
    }
    /* pixel centres at 16*p+8 inside [min,max]: p >= (min-8)/16 (ceil), p <= (max-8)/16 (floor) */
    int px0 = (minx - 8 + 15) >> 4, px1 = (maxx - 8) >> 4;
    int py0 = (miny - 8 + 15) >> 4, py1 = (maxy - 8) >> 4;
    if (px0 < 0) px0 = 0; if (py0 < 0) py0 = 0;
    if (px1 > W - 1) px1 = W - 1; if (py1 > H - 1) py1 = H - 1;
    if (px0 > px1 || py0 > py1) return;
    int64_t A[3], Bc[3], C[3];
    for (int i = 0; i < 3; i++) {
        int a = (i + 1) % 3, bb = (i + 2) % 3;
        A[i] = (int64_t)s.y[a] - s.y[bb];
        Bc[i] = (int64_t)s.x[bb] - s.x[a];
        C[i] = -Bc[i] * s.y[a] - A[i] * s.x[a];
    }
    for (int py = py0; py <= py1; py++) {
        for (int px = px0; px <= px1; px++) {
            int64_t cx = 16 * (int64_t)px + 8, cy = 16 * (int64_t)py + 8;
            int inside = 1;
            for (int i = 0; i < 3 && inside; i++) {
                int64_t E = A[i] * cx + Bc[i] * cy + C[i];
                if (E < 0) inside = 0;
                else if (E == 0 && !(A[i] > 0 || (A[i] == 0 && Bc[i] > 0))) inside = 0;
            }
            if (!inside) continue;
            zplane_t zp = depth_plane(p0, p1, p2, &s, area, px & ~7, py & ~7);
            float zt = fmaf(zp.gx, (float)(px & 7), fmaf(zp.gy, (float)(py & 7), zp.zwc));
            if (!(zt >= -1.0f && zt <= 1.0f)) continue;
            uint64_t key = ((uint64_t)f2ord(zt) << 32) | (uint32_t)t;
            uint64_t* dst = &vis[(size_t)py * W + px];
            if (key < *dst) *dst = key;
        }
    }
}

/* pos [B,V,4] f32, tri [F,3] i32 -> rast [B,H,W,4], rast_db [B,H,W,4] (may be NULL) */
int oracle_rasterize(const float* pos, const int32_t* tri, int B, int V, int F, int H, int W,
                     float* rast, float* rast_db) {
    if (!pos || !tri || !rast || B < 0 || V <= 0 || F < 0 || H <= 0 || W <= 0) return -1;
    const float xs = 2.0f / (float)W, xo = 1.0f / (float)W - 1.0f;
    const float ys = 2.0f / (float)H, yo = 1.0f / (float)H - 1.0f;
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; b++) {
        const float* P = pos + (size_t)b * V * 4;
        uint64_t* vis = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)H * W);
        if (!vis) { err = -2; continue; }
        memset(vis, 0xff, sizeof(uint64_t) * (size_t)H * W);
        for (int t = 0; t < F; t++) {
            int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
            if ((unsigned)i0 >= (unsigned)V || (unsigned)i1 >= (unsigned)V || (unsigned)i2 >= (unsigned)V) continue;
            float q[2][3][4];
            int np = clip_near(P + 4 * i0, P + 4 * i1, P + 4 * i2, q);
            for (int k = 0; k < np; k++) raster_piece(q[k][0], q[k][1], q[k][2], t, H, W, vis);
        }
        /* shading pass */
        for (int py = 0; py < H; py++) {
            for (int px = 0; px < W; px++) {
                size_t pidx = ((size_t)b * H + py) * W + px;
                float* o = rast + 4 * pidx;
                float* d = rast_db ? rast_db + 4 * pidx : NULL;
                uint64_t key = vis[(size_t)py * W + px];
                if (key == ~(uint64_t)0) {
                    o[0] = o[1] = o[2] = o[3] = 0.0f;
                    if (d) d[0] = d[1] = d[2] = d[3] = 0.0f;
                    continue;
                }
                int t = (int)(uint32_t)key;
                const float *p0 = P + 4 * tri[3 * t], *p1 = P + 4 * tri[3 * t + 1], *p2 = P + 4 * tri[3 * t + 2];
                float fx = fmaf(xs, (float)px, xo), fy = fmaf(ys, (float)py, yo);
                frag_t fr = shade_frag(p0, p1, p2, fx, fy);
                o[0] = fr.b0; o[1] = fr.b1; o[2] = fr.zw; o[3] = (float)(t + 1);
                if (d) {
                    float dfxdx = xs * fr.iw, dfydy = ys * fr.iw;
                    float da0dx = fmaf(p2[1], p1[3], -(p1[1] * p2[3]));
                    float da0dy = fmaf(p1[0], p2[3], -(p2[0] * p1[3]));
                    float da1dx = fmaf(p0[1], p2[3], -(p2[1] * p0[3]));
                    float da1dy = fmaf(p2[0], p0[3], -(p0[0] * p2[3]));
                    float da2dx = fmaf(p1[1], p0[3], -(p0[1] * p1[3]));
                    float da2dy = fmaf(p0[0], p1[3], -(p1[0] * p0[3]));
                    float datdx = (da0dx + da1dx) + da2dx;
                    float datdy = (da0dy + da1dy) + da2dy;
                    d[0] = dfxdx * fmaf(fr.b0, datdx, -da0dx);
                    d[1] = dfydy * fmaf(fr.b0, datdy, -da0dy);
                    d[2] = dfxdx * fmaf(fr.b1, datdx, -da1dx);
                    d[3] = dfydy * fmaf(fr.b1, datdy, -da1dy);
                }
            }
        }
        free(vis);
    }
    return err;
}

/* attr [AB,V,A] (AB == 1 broadcasts, else AB == B), rast [B,H,W,4], tri [F,3]
 * out [B,H,W,A]; if rast_db && out_da: out_da [B,H,W,2A] = (da/dX, da/dY) per attribute.
 *   out = fma(b0, a0, fma(b1, a1, b2*a2)),  b2 = (1 - b0) - b1
 *   da/dX = fma(dudx, a0 - a2, dvdx * (a1 - a2))                                       */
int oracle_interpolate(const float* attr, int AB, const float* rast, const int32_t* tri,
                       const float* rast_db, int B, int H, int W, int V, int F, int A,
                       float* out, float* out_da) {
    if (!attr || !rast || !tri || !out || (AB != 1 && AB != B) || A <= 0) return -1;
    size_t npix = (size_t)B * H * W;
#pragma omp parallel for schedule(static)
    for (long long pi = 0; pi < (long long)npix; pi++) {
        const float* r = rast + 4 * pi;
        float* o = out + (size_t)A * pi;
        float* od = (rast_db && out_da) ? out_da + 2 * (size_t)A * pi : NULL;
        int t = (int)r[3] - 1;
        if (t < 0 || t >= F) {
            for (int k = 0; k < A; k++) o[k] = 0.0f;
            if (od) for (int k = 0; k < 2 * A; k++) od[k] = 0.0f;
            continue;
        }
        int b = (int)(pi / ((size_t)H * W));
        const float* base = attr + (AB == 1 ? 0 : (size_t)b * V * A);
        const float* a0 = base + (size_t)tri[3 * t] * A;
        const float* a1 = base + (size_t)tri[3 * t + 1] * A;
        const float* a2 = base + (size_t)tri[3 * t + 2] * A;
        float b0 = r[0], b1 = r[1], b2 = (1.0f - b0) - b1;
        for (int k = 0; k < A; k++) o[k] = fmaf(b0, a0[k], fmaf(b1, a1[k], b2 * a2[k]));
        if (od) {
            const float* d = rast_db + 4 * pi;
            for (int k = 0; k < A; k++) {
                float e0 = a0[k] - a2[k], e1 = a1[k] - a2[k];
                od[2 * k] = fmaf(d[0], e0, d[2] * e1);
                od[2 * k + 1] = fmaf(d[1], e0, d[3] * e1);
            }
        }
    }
    return 0;
}
