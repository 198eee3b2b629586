// Attribute interpolation (forward / backward) and rasterizer backward for gfx950.
//
// Replaces dr.interpolate and the backward of dr.rasterize of the reference
// (vhap/util/render_nvdiffrast.py:254,384,389; autograd replays them at tracker.py:1434).
// One thread per pixel; 16-byte coalesced reads of rast / rast_db; gradients to vertices go through
// hardware fp32 atomics (global_atomic_add_f32, -munsafe-fp-atomics) -- float order is therefore
// not deterministic, tests compare within 1e-4 relative.
#include "common.h"
#include "gbuffer_tile.h"

namespace {

constexpr int MAX_ATTR = 16;

__global__ __launch_bounds__(256) void interp_fwd_kernel(const float* __restrict__ attr, int AB,
                                                         const float4* __restrict__ rast, const int* __restrict__ tri,
                                                         const float4* __restrict__ rast_db, long long npix, int HW, int V,
                                                         int F, int A, float* __restrict__ out, float* __restrict__ out_da) {
    const long long pi = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pi >= npix) return;
    const float4 r = rast[pi];
    float* o = out + (size_t)A * pi;
    float* od = out_da ? out_da + 2 * (size_t)A * pi : nullptr;
    const int t = (int)r.w - 1;
    if (t < 0 || t >= F) {
        for (int k = 0; k < A; k++) o[k] = 0.0f;
        if (od) for (int k = 0; k < 2 * A; k++) od[k] = 0.0f;
        return;
    }
    const int b = (int)(pi / HW);
    const float* base = attr + (AB == 1 ? 0 : (size_t)b * V * A);
    const float* a0 = base + (size_t)tri[3 * t] * A;
    const float* a1 = base + (size_t)tri[3 * t + 1] * A;
    const float* a2 = base + (size_t)tri[3 * t + 2] * A;
    const float b0 = r.x, b1 = r.y, b2 = (1.0f - b0) - b1;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (od) d = rast_db[pi];
    for (int k = 0; k < A; k++) {
        const float x0 = a0[k], x1 = a1[k], x2 = a2[k];
        o[k] = __fmaf_rn(b0, x0, __fmaf_rn(b1, x1, b2 * x2));
        if (od) {
            const float e0 = x0 - x2, e1 = x1 - x2;
            od[2 * k] = __fmaf_rn(d.x, e0, d.z * e1);
            od[2 * k + 1] = __fmaf_rn(d.y, e0, d.w * e1);
        }
    }
}

__global__ __launch_bounds__(256) void interp_bwd_kernel(const float* __restrict__ attr, int AB,
                                                         const float4* __restrict__ rast, const int* __restrict__ tri,
                                                         const float4* __restrict__ rast_db, const float* __restrict__ d_out,
                                                         const float* __restrict__ d_out_da, long long npix, int HW, int V,
                                                         int F, int A, float* __restrict__ d_attr,
                                                         float4* __restrict__ d_rast, float4* __restrict__ d_rast_db) {
    const long long pi = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pi >= npix) return;
    const float4 r = rast[pi];
    const int t = (int)r.w - 1;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < 0 || t >= F) {
        if (d_rast) d_rast[pi] = z4;
        if (d_rast_db) d_rast_db[pi] = z4;
        return;
    }
    const int b = (int)(pi / HW);
    const size_t boff = AB == 1 ? 0 : (size_t)b * V * A;
    const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
    const float* a0 = attr + boff + (size_t)i0 * A;
    const float* a1 = attr + boff + (size_t)i1 * A;
    const float* a2 = attr + boff + (size_t)i2 * A;
    float* g0 = d_attr ? d_attr + boff + (size_t)i0 * A : nullptr;
    float* g1 = d_attr ? d_attr + boff + (size_t)i1 * A : nullptr;
    float* g2 = d_attr ? d_attr + boff + (size_t)i2 * A : nullptr;
    const float b0 = r.x, b1 = r.y, b2 = (1.0f - b0) - b1;
    const bool has_da = d_out_da != nullptr && rast_db != nullptr;
    float4 d = z4;
    if (has_da) d = rast_db[pi];
    const float* go = d_out + (size_t)A * pi;
    const float* gd = has_da ? d_out_da + 2 * (size_t)A * pi : nullptr;
    float gb0 = 0.f, gb1 = 0.f;
    float4 gdb = z4;
    for (int k = 0; k < A; k++) {
        const float x0 = a0[k], x1 = a1[k], x2 = a2[k];
        const float e0 = x0 - x2, e1 = x1 - x2;
        const float g = go[k];
        gb0 += g * e0;
        gb1 += g * e1;
        float ga0 = b0 * g, ga1 = b1 * g, ga2 = b2 * g;
        if (has_da) {
            const float gx = gd[2 * k], gy = gd[2 * k + 1];
            gdb.x += gx * e0; gdb.z += gx * e1;
            gdb.y += gy * e0; gdb.w += gy * e1;
            const float ge0 = gx * d.x + gy * d.y, ge1 = gx * d.z + gy * d.w;
            ga0 += ge0; ga1 += ge1; ga2 -= ge0 + ge1;
        }
        if (g0) {
            if (ga0 != 0.f) atomicAdd(&g0[k], ga0);
            if (ga1 != 0.f) atomicAdd(&g1[k], ga1);
            if (ga2 != 0.f) atomicAdd(&g2[k], ga2);
        }
    }
    if (d_rast) d_rast[pi] = make_float4(gb0, gb1, 0.f, 0.f);
    if (d_rast_db) d_rast_db[pi] = gdb;
}

// d(u, v, du/dX, du/dY, dv/dX, dv/dY) / d(clip-space vertex positions), see shade_frag() in raster.hip
__global__ __launch_bounds__(256) void raster_bwd_kernel(const float4* __restrict__ pos, const int* __restrict__ tri,
                                                         const float4* __restrict__ rast, const float4* __restrict__ d_rast,
                                                         const float4* __restrict__ d_rast_db, long long npix, int H, int W,
                                                         int V, int F, float* __restrict__ d_pos) {
    const long long pi = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pi >= npix) return;
    const float4 r = rast[pi];
    const int t = (int)r.w - 1;
    if (t < 0 || t >= F) return;
    float4 g = d_rast[pi];
    float4 gd = d_rast_db ? d_rast_db[pi] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.x == 0.f && g.y == 0.f && gd.x == 0.f && gd.y == 0.f && gd.z == 0.f && gd.w == 0.f) return;
    const int HW = H * W;
    const int b = (int)(pi / HW);
    const int rem = (int)(pi - (long long)b * HW);
    const int py = rem / W, px = rem - py * W;
    const float xs = 2.0f / (float)W, xo = 1.0f / (float)W - 1.0f;
    const float ys = 2.0f / (float)H, yo = 1.0f / (float)H - 1.0f;
    const float fx = xs * (float)px + xo, fy = ys * (float)py + yo;
    const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
    const float4* P = pos + (size_t)b * V;
    const float4 p0 = P[i0], p1 = P[i1], p2 = P[i2];
    const float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
    const float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
    const float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
    const float a0 = p1x * p2y - p1y * p2x;
    const float a1 = p2x * p0y - p2y * p0x;
    const float a2 = p0x * p1y - p0y * p1x;
    const float at = a0 + a1 + a2;
    if (!(fabsf(at) > 0.0f)) return;
    const float iw = 1.0f / at;
    const float r0 = a0 * iw, r1 = a1 * iw;          // unclamped barycentrics
    const float b0 = r.x, b1 = r.y;                  // clamped (as output by the forward)
    // gradient arriving at the clamped barycentrics (direct + through the pixel differentials)
    const float X0 = p2.y * p1.w - p1.y * p2.w, Y0 = p1.x * p2.w - p2.x * p1.w;
    const float X1 = p0.y * p2.w - p2.y * p0.w, Y1 = p2.x * p0.w - p0.x * p2.w;
    const float X2 = p1.y * p0.w - p0.y * p1.w, Y2 = p0.x * p1.w - p1.x * p0.w;
    const float Tx = X0 + X1 + X2, Ty = Y0 + Y1 + Y2;
    float g0 = g.x + xs * iw * Tx * gd.x + ys * iw * Ty * gd.y;
    float g1 = g.y + xs * iw * Tx * gd.z + ys * iw * Ty * gd.w;
    if (!(r0 >= 0.0f && r0 <= 1.0f)) g0 = 0.0f;      // clamp() passes no gradient outside [0,1]
    if (!(r1 >= 0.0f && r1 <= 1.0f)) g1 = 0.0f;
    const float giw = xs * (b0 * Tx - X0) * gd.x + ys * (b0 * Ty - Y0) * gd.y + xs * (b1 * Tx - X1) * gd.z +
                      ys * (b1 * Ty - Y1) * gd.w;
    const float s = g0 * r0 + g1 * r1;
    const float gat = -iw * iw * giw;
    const float ga0 = (g0 - s) * iw + gat, ga1 = (g1 - s) * iw + gat, ga2 = (-s) * iw + gat;
    // a0 = p1x*p2y - p1y*p2x ; a1 = p2x*p0y - p2y*p0x ; a2 = p0x*p1y - p0y*p1x
    const float gp0x = -p2y * ga1 + p1y * ga2, gp0y = p2x * ga1 - p1x * ga2;
    const float gp1x = p2y * ga0 - p0y * ga2, gp1y = -p2x * ga0 + p0x * ga2;
    const float gp2x = -p1y * ga0 + p0y * ga1, gp2y = p1x * ga0 - p0x * ga1;
    float d0x = gp0x, d0y = gp0y, d0w = -fx * gp0x - fy * gp0y;
    float d1x = gp1x, d1y = gp1y, d1w = -fx * gp1x - fy * gp1y;
    float d2x = gp2x, d2y = gp2y, d2w = -fx * gp2x - fy * gp2y;
    // through X*, Y* (the raw-coordinate terms of the pixel differentials)
    const float cx = xs * iw, cy = ys * iw;
    const float sx = b0 * gd.x + b1 * gd.z, sy = b0 * gd.y + b1 * gd.w;
    const float gX0 = cx * (sx - gd.x), gX1 = cx * (sx - gd.z), gX2 = cx * sx;
    const float gY0 = cy * (sy - gd.y), gY1 = cy * (sy - gd.w), gY2 = cy * sy;
    d2y += p1.w * gX0; d1w += p2.y * gX0; d1y -= p2.w * gX0; d2w -= p1.y * gX0;
    d0y += p2.w * gX1; d2w += p0.y * gX1; d2y -= p0.w * gX1; d0w -= p2.y * gX1;
    d1y += p0.w * gX2; d0w += p1.y * gX2; d0y -= p1.w * gX2; d1w -= p0.y * gX2;
    d1x += p2.w * gY0; d2w += p1.x * gY0; d2x -= p1.w * gY0; d1w -= p2.x * gY0;
    d2x += p0.w * gY1; d0w += p2.x * gY1; d0x -= p2.w * gY1; d2w -= p0.x * gY1;
    d0x += p1.w * gY2; d1w += p0.x * gY2; d1x -= p0.w * gY2; d0w -= p1.x * gY2;
    float* D = d_pos + (size_t)b * V * 4;
    atomicAdd(&D[4 * i0 + 0], d0x); atomicAdd(&D[4 * i0 + 1], d0y); atomicAdd(&D[4 * i0 + 3], d0w);
    atomicAdd(&D[4 * i1 + 0], d1x); atomicAdd(&D[4 * i1 + 1], d1y); atomicAdd(&D[4 * i1 + 3], d1w);
    atomicAdd(&D[4 * i2 + 0], d2x); atomicAdd(&D[4 * i2 + 1], d2y); atomicAdd(&D[4 * i2 + 3], d2w);
}

}  // namespace

extern "C" int vhap_interp_fwd(const float* attr, int AB, const float* rast, const int32_t* tri, const float* rast_db, int B,
                               int H, int W, int V, int F, int A, float* out, float* out_da, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!attr || !rast || !tri || !out) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || V <= 0 || F <= 0 || A <= 0 || A > MAX_ATTR || (AB != 1 && AB != B)) return VHAP_E_BADDIM;
    if ((out_da != nullptr) != (rast_db != nullptr) && out_da) return VHAP_E_NULLPTR;
    const long long npix = (long long)B * H * W;
    interp_fwd_kernel<<<vhap_cdiv(npix, 256), 256, 0, vhap_stream(stream)>>>(
        attr, AB, reinterpret_cast<const float4*>(rast), tri, reinterpret_cast<const float4*>(rast_db), npix, H * W, V, F, A, out,
        rast_db ? out_da : nullptr);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_interp_bwd(const float* attr, int AB, const float* rast, const int32_t* tri, const float* rast_db,
                               const float* d_out, const float* d_out_da, int B, int H, int W, int V, int F, int A,
                               float* d_attr, float* d_rast, float* d_rast_db, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!attr || !rast || !tri || !d_out) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || V <= 0 || F <= 0 || A <= 0 || A > MAX_ATTR || (AB != 1 && AB != B)) return VHAP_E_BADDIM;
    const long long npix = (long long)B * H * W;
    interp_bwd_kernel<<<vhap_cdiv(npix, 256), 256, 0, vhap_stream(stream)>>>(
        attr, AB, reinterpret_cast<const float4*>(rast), tri, reinterpret_cast<const float4*>(rast_db), d_out, d_out_da, npix,
        H * W, V, F, A, d_attr, reinterpret_cast<float4*>(d_rast), reinterpret_cast<float4*>(d_rast_db));
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_raster_bwd(const float* pos, const int32_t* tri, const float* rast, const float* d_rast,
                               const float* d_rast_db, int B, int V, int F, int H, int W, float* d_pos, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!pos || !tri || !rast || !d_rast || !d_pos) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || V <= 0 || F <= 0) return VHAP_E_BADDIM;
    const long long npix = (long long)B * H * W;
    raster_bwd_kernel<<<vhap_cdiv(npix, 256), 256, 0, vhap_stream(stream)>>>(
        reinterpret_cast<const float4*>(pos), tri, reinterpret_cast<const float4*>(rast), reinterpret_cast<const float4*>(d_rast),
        reinterpret_cast<const float4*>(d_rast_db), npix, H, W, V, F, d_pos);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Triangle-parallel backward of the fused G-buffer pass (rasterize + interpolate(normal) + interpolate(uv,'all')).
//
// The per-pixel backward kernels above issue 9-18 float atomics per covered pixel (12 M per 16x512^2 batch,
// ~2.6 ms).  Here one thread owns one (frame, triangle): it walks the triangle's pixel bounding box (same
// snapping rules as the rasteriser), picks the pixels the triangle actually won (rast.w == id + 1), chains
//   d_normal, d_texc, d_texd -> d(u, v), d(du/dX ..) -> d(clip positions), d(vertex normals)
// per pixel in registers, and issues ONE set of atomics per triangle vertex at the end (18 per triangle instead
// of per pixel).  Nothing per-pixel is materialised (no d_rast / d_rast_db buffers).
#include "raster_common.h"

namespace {

// 16 lanes cooperate on one (frame, triangle): lane s visits bbox pixels s, s+16, ...; a 4-step butterfly
// reduces the 18 accumulators inside the 16-lane row; lanes 0..8 then issue the atomics.
constexpr int GB_LANES = 16;

__global__ __launch_bounds__(256) void gbuffer_bwd_kernel(const float4* __restrict__ pos, const int* __restrict__ tri,
                                                          const float* __restrict__ vnormal, const float2* __restrict__ uv,
                                                          const int* __restrict__ tri_uv, const float4* __restrict__ rast,
                                                          const float* __restrict__ d_normal, const float2* d_texc,
                                                          const float4* d_texd, const float4* __restrict__ d_rast,
                                                          const float4* __restrict__ d_db, const unsigned char* __restrict__ uv_nograd,
                                                          int B, int V, int F, int H, int W, float* __restrict__ d_pos,
                                                          float* __restrict__ d_vnormal) {
    const long long gid = ((long long)blockIdx.x * 256 + threadIdx.x) / GB_LANES;
    const int sub = threadIdx.x & (GB_LANES - 1);
    if (gid >= (long long)B * F) return;   // whole 16-lane group exits together
    const int b = (int)(gid / F), t = (int)(gid - (long long)b * F);
    const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
    if ((unsigned)i0 >= (unsigned)V || (unsigned)i1 >= (unsigned)V || (unsigned)i2 >= (unsigned)V) return;
    const float4* P = pos + (size_t)b * V;
    const float4 p0 = P[i0], p1 = P[i1], p2 = P[i2];
    int px0, px1, py0, py1;
    if (!tri_cover_bbox(p0, p1, p2, H, W, px0, px1, py0, py1)) return;     // (union of the pieces of a triangle cut by the near plane)
    const float xs = 2.0f / (float)W, xo = 1.0f / (float)W - 1.0f;
    const float ys = 2.0f / (float)H, yo = 1.0f / (float)H - 1.0f;
    const float X0 = p2.y * p1.w - p1.y * p2.w, Y0 = p1.x * p2.w - p2.x * p1.w;
    const float X1 = p0.y * p2.w - p2.y * p0.w, Y1 = p2.x * p0.w - p0.x * p2.w;
    const float X2 = p1.y * p0.w - p0.y * p1.w, Y2 = p0.x * p1.w - p1.x * p0.w;
    const float Tx = X0 + X1 + X2, Ty = Y0 + Y1 + Y2;
    float nd0[3] = {0.f, 0.f, 0.f}, nd1[3] = {0.f, 0.f, 0.f};
    if (vnormal) {
        const float* N = vnormal + (size_t)b * V * 3;
#pragma unroll
        for (int k = 0; k < 3; k++) { nd0[k] = N[3 * i0 + k] - N[3 * i2 + k]; nd1[k] = N[3 * i1 + k] - N[3 * i2 + k]; }
    }
    float2 e0 = make_float2(0.f, 0.f), e1 = e0;
    if (uv && uv_nograd && uv_nograd[t]) d_texc = nullptr;   // texc detached on this face (its screen-space derivatives texd are not)
    if (uv) {
        const float2 u0 = uv[tri_uv[3 * t]], u1 = uv[tri_uv[3 * t + 1]], u2 = uv[tri_uv[3 * t + 2]];
        e0 = make_float2(u0.x - u2.x, u0.y - u2.y);
        e1 = make_float2(u1.x - u2.x, u1.y - u2.y);
    }
    float acc[18];   // [0..8] = d_pos (vertex-major: x, y, w), [9..17] = d_vnormal (vertex-major)
#pragma unroll
    for (int k = 0; k < 18; k++) acc[k] = 0.f;
    const float idf = (float)(t + 1);
    const int bw = px1 - px0 + 1, npx = bw * (py1 - py0 + 1);
    for (int k = sub; k < npx; k += GB_LANES) {
        const int ry = k / bw, px = px0 + (k - ry * bw), py = py0 + ry;
        const size_t pi = ((size_t)b * H + py) * W + px;
        const float4 r = rast[pi];
        if (r.w != idf) continue;
        const float b0 = r.x, b1 = r.y, b2 = (1.0f - b0) - b1;
        float g0 = 0.f, g1 = 0.f;
        float4 gd = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d_rast) { const float4 q = d_rast[pi]; g0 = q.x; g1 = q.y; }
        if (d_db) gd = d_db[pi];
        if (d_normal) {
            const float* gn = d_normal + 3 * pi;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float v = gn[c];
                acc[9 + c] += b0 * v; acc[12 + c] += b1 * v; acc[15 + c] += b2 * v;
                g0 += v * nd0[c]; g1 += v * nd1[c];
            }
        }
        if (d_texc) {
            const float2 gt = d_texc[pi];
            g0 += gt.x * e0.x + gt.y * e0.y;
            g1 += gt.x * e1.x + gt.y * e1.y;
        }
        if (d_texd) {
            const float4 q = d_texd[pi];   // grads of (du/dX, du/dY, dv/dX, dv/dY) of the texture coordinate
            gd.x += q.x * e0.x + q.z * e0.y; gd.z += q.x * e1.x + q.z * e1.y;
            gd.y += q.y * e0.x + q.w * e0.y; gd.w += q.y * e1.x + q.w * e1.y;
        }
        // ---- rasterizer backward (same algebra as raster_bwd_kernel) ----
        const float fx = xs * (float)px + xo, fy = ys * (float)py + yo;
        const float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
        const float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
        const float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
        const float a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x;
        const float at = a0 + a1 + a2;
        if (!(fabsf(at) > 0.0f)) continue;
        const float iw = 1.0f / at;
        const float r0 = a0 * iw, r1 = a1 * iw;
        float G0 = g0 + xs * iw * Tx * gd.x + ys * iw * Ty * gd.y;
        float G1 = g1 + xs * iw * Tx * gd.z + ys * iw * Ty * gd.w;
        if (!(r0 >= 0.0f && r0 <= 1.0f)) G0 = 0.0f;
        if (!(r1 >= 0.0f && r1 <= 1.0f)) G1 = 0.0f;
        const float giw = xs * (b0 * Tx - X0) * gd.x + ys * (b0 * Ty - Y0) * gd.y + xs * (b1 * Tx - X1) * gd.z +
                          ys * (b1 * Ty - Y1) * gd.w;
        const float s = G0 * r0 + G1 * r1;
        const float gat = -iw * iw * giw;
        const float ga0 = (G0 - s) * iw + gat, ga1 = (G1 - s) * iw + gat, ga2 = (-s) * iw + gat;
        const float gp0x = -p2y * ga1 + p1y * ga2, gp0y = p2x * ga1 - p1x * ga2;
        const float gp1x = p2y * ga0 - p0y * ga2, gp1y = -p2x * ga0 + p0x * ga2;
        const float gp2x = -p1y * ga0 + p0y * ga1, gp2y = p1x * ga0 - p0x * ga1;
        acc[0] += gp0x; acc[1] += gp0y; acc[2] += -fx * gp0x - fy * gp0y;
        acc[3] += gp1x; acc[4] += gp1y; acc[5] += -fx * gp1x - fy * gp1y;
        acc[6] += gp2x; acc[7] += gp2y; acc[8] += -fx * gp2x - fy * gp2y;
        const float cx = xs * iw, cy = ys * iw;
        const float sxg = b0 * gd.x + b1 * gd.z, syg = b0 * gd.y + b1 * gd.w;
        const float gX0 = cx * (sxg - gd.x), gX1 = cx * (sxg - gd.z), gX2 = cx * sxg;
        const float gY0 = cy * (syg - gd.y), gY1 = cy * (syg - gd.w), gY2 = cy * syg;
        acc[7] += p1.w * gX0; acc[5] += p2.y * gX0; acc[4] -= p2.w * gX0; acc[8] -= p1.y * gX0;
        acc[1] += p2.w * gX1; acc[8] += p0.y * gX1; acc[7] -= p0.w * gX1; acc[2] -= p2.y * gX1;
        acc[4] += p0.w * gX2; acc[2] += p1.y * gX2; acc[1] -= p1.w * gX2; acc[5] -= p0.y * gX2;
        acc[3] += p2.w * gY0; acc[8] += p1.x * gY0; acc[6] -= p1.w * gY0; acc[5] -= p2.x * gY0;
        acc[6] += p0.w * gY1; acc[2] += p2.x * gY1; acc[0] -= p2.w * gY1; acc[8] -= p0.x * gY1;
        acc[0] += p1.w * gY2; acc[5] += p0.x * gY2; acc[3] -= p0.w * gY2; acc[2] -= p1.x * gY2;
    }
    // butterfly over the 16-lane group (every lane ends up with the full sums)
#pragma unroll
    for (int k = 0; k < 18; k++) {
        float v = acc[k];
#pragma unroll
        for (int o = GB_LANES / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, GB_LANES);
        acc[k] = v;
    }
    if (sub < 9) {
        const int vtx = sub / 3, c = sub - 3 * vtx;
        const int vi = vtx == 0 ? i0 : (vtx == 1 ? i1 : i2);
        float vp = 0.f, vn = 0.f;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            if (k == sub) { vp = acc[k]; vn = acc[9 + k]; }
        }
        if (d_pos && vp != 0.f) atomicAdd(&d_pos[((size_t)b * V + vi) * 4 + (c == 2 ? 3 : c)], vp);
        if (d_vnormal && d_normal && vn != 0.f) atomicAdd(&d_vnormal[((size_t)b * V + vi) * 3 + c], vn);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Pixel-parallel variant with LDS aggregation (gbuffer_tile.h).  A workgroup owns a 16x16 pixel tile of one frame; no lane ever walks
// pixels it does not own -- the triangle-parallel kernel above spends most of its time on bounding-box pixels won by other triangles.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GT * GT) void gbuffer_bwd_tiled_kernel(const float4* __restrict__ pos, const int* __restrict__ tri,
                                                                    const float* __restrict__ vnormal, const float2* __restrict__ uv,
                                                                    const int* __restrict__ tri_uv, const float4* __restrict__ rast,
                                                                    const float* __restrict__ d_normal, const float2* __restrict__ d_texc,
                                                                    const float4* __restrict__ d_texd, const float4* __restrict__ d_rast,
                                                                    const float4* __restrict__ d_db, const unsigned char* __restrict__ uv_nograd,
                                                                    int V, int F, int H, int W, float* __restrict__ d_pos,
                                                                    float* __restrict__ d_vnormal, int dbg) {
    __shared__ GbTile S;
    const int tid = threadIdx.x;
    const int px = blockIdx.x * GT + (tid & (GT - 1)), py = blockIdx.y * GT + (tid >> 4), b = blockIdx.z;
    const bool inside = px < W && py < H;
    const size_t pi = ((size_t)b * H + (inside ? py : 0)) * W + (inside ? px : 0);
    const float4 r = inside ? rast[pi] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int t = (int)r.w - 1;
    const bool cov = inside && t >= 0 && t < F;
    if (__syncthreads_or(cov ? 1 : 0) == 0) return;       // background tile
    gb_tile_init(S);
    __syncthreads();
    float acc[18];
#pragma unroll
    for (int k = 0; k < 18; k++) acc[k] = 0.f;
    int i0 = 0, i1 = 0, i2 = 0;
    bool have = false;
    if (cov) {
        // A covered wave's life is a chain of dependent gathers (75 % of its cycles at s_waitcnt, profiles/r04_call22_step_sq_pmc.json), so
        // the loads are ISSUED in two batches -- everything addressed by the pixel or the triangle id, then everything addressed by a
        // vertex -- and consumed afterwards.  Written in the order of use, each optional input sat behind its own uniform branch and the
        // compiler waited for them one after the other: nine round trips where three do.
        const bool want_n = d_normal && vnormal, want_uv = uv && (d_texc || d_texd);
        i0 = tri[3 * t]; i1 = tri[3 * t + 1]; i2 = tri[3 * t + 2];
        // (an absent optional input is read from a valid stand-in address and discarded by a uniform select: a branch around the load
        // ends in a join, where the compiler waits for it)
        const int* a_tuv = want_uv ? tri_uv : tri;
        const float4* a_rast = d_rast ? d_rast : rast;
        const float4* a_db = d_db ? d_db : rast;
        const float* a_n = want_n ? d_normal : reinterpret_cast<const float*>(rast);
        const float2* a_texc = want_uv && d_texc ? d_texc : reinterpret_cast<const float2*>(rast);
        const float4* a_texd = want_uv && d_texd ? d_texd : rast;
        const unsigned char* a_ng = want_uv && d_texc && uv_nograd ? uv_nograd : reinterpret_cast<const unsigned char*>(tri);
        const int j0 = a_tuv[3 * t], j1 = a_tuv[3 * t + 1], j2 = a_tuv[3 * t + 2];
        const float4 l_rast = a_rast[pi], l_db = a_db[pi], l_texd = a_texd[pi];
        const float2 l_texc = a_texc[pi];
        const float l_n0 = a_n[3 * pi], l_n1 = a_n[3 * pi + 1], l_n2 = a_n[3 * pi + 2];
        const unsigned char l_ng = a_ng[t];
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 q_rast = d_rast ? l_rast : zero4, q_db = d_db ? l_db : zero4, q_texd = want_uv && d_texd ? l_texd : zero4;
        const float2 q_texc = want_uv && d_texc ? l_texc : make_float2(0.f, 0.f);
        const float gn0 = want_n ? l_n0 : 0.f, gn1 = want_n ? l_n1 : 0.f, gn2 = want_n ? l_n2 : 0.f;
        const unsigned char nograd = want_uv && d_texc && uv_nograd ? l_ng : (unsigned char)0;
        if ((unsigned)i0 < (unsigned)V && (unsigned)i1 < (unsigned)V && (unsigned)i2 < (unsigned)V) {
            have = true;
            const float4* P = pos + (size_t)b * V;
            const float4 p0 = P[i0], p1 = P[i1], p2 = P[i2];
            float n0[3], n1[3], n2[3];
            const float* N = want_n ? vnormal + (size_t)b * V * 3 : reinterpret_cast<const float*>(P);      // (stand-in: 3 i + c < 4 V)
#pragma unroll
            for (int c = 0; c < 3; c++) { n0[c] = N[3 * i0 + c]; n1[c] = N[3 * i1 + c]; n2[c] = N[3 * i2 + c]; }
            const float2* U = want_uv ? uv : reinterpret_cast<const float2*>(P);                             // (stand-in: j = a vertex index here)
            const float2 u0 = U[j0], u1 = U[j1], u2 = U[j2];
            const float b0 = r.x, b1 = r.y, b2 = (1.0f - b0) - b1;
            float g0 = q_rast.x, g1 = q_rast.y;
            float4 gd = q_db;
            if (want_n) {
                const float gnv[3] = {gn0, gn1, gn2};
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float v = gnv[c];
                    acc[9 + c] = b0 * v; acc[12 + c] = b1 * v; acc[15 + c] = b2 * v;
                    g0 += v * (n0[c] - n2[c]); g1 += v * (n1[c] - n2[c]);
                }
            }
            if (want_uv) {
                const float2 e0 = make_float2(u0.x - u2.x, u0.y - u2.y), e1 = make_float2(u1.x - u2.x, u1.y - u2.y);
                if (d_texc && !nograd) {
                    g0 += q_texc.x * e0.x + q_texc.y * e0.y;
                    g1 += q_texc.x * e1.x + q_texc.y * e1.y;
                }
                if (d_texd) {
                    const float4 q = q_texd;
                    gd.x += q.x * e0.x + q.z * e0.y; gd.z += q.x * e1.x + q.z * e1.y;
                    gd.y += q.y * e0.x + q.w * e0.y; gd.w += q.y * e1.x + q.w * e1.y;
                }
            }
            gb_chain(p0, p1, p2, b0, b1, px, py, H, W, g0, g1, gd, acc);
        }
    }
    gb_tile_commit(S, acc, have, i0, i1, i2, b, V, d_pos, d_vnormal, dbg);
}

}  // namespace

extern "C" int vhap_gbuffer_bwd(const float* pos, const int32_t* tri, const float* vnormal, const float* uv, const int32_t* tri_uv,
                                const float* rast, const float* d_normal, const float* d_texc, const float* d_texd,
                                const float* d_rast, const float* d_rast_db, const uint8_t* uv_nograd_faces, int B, int V, int F, int H,
                                int W, float* d_pos, float* d_vnormal, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!pos || !tri || !rast) return VHAP_E_NULLPTR;
    if ((d_normal && !vnormal) || ((d_texc || d_texd) && (!uv || !tri_uv))) return VHAP_E_NULLPTR;
    if (B <= 0 || V <= 0 || F <= 0 || H <= 0 || W <= 0 || (long long)B * F >= (1ll << 31)) return VHAP_E_BADDIM;
    if (!(vhap_g_debug_flags & 64)) {          // (flag 64: A/B switch to the triangle-parallel kernel)
        gbuffer_bwd_tiled_kernel<<<dim3(vhap_cdiv(W, GT), vhap_cdiv(H, GT), B), GT * GT, 0, vhap_stream(stream)>>>(
            reinterpret_cast<const float4*>(pos), tri, vnormal, reinterpret_cast<const float2*>(uv), tri_uv,
            reinterpret_cast<const float4*>(rast), d_normal, reinterpret_cast<const float2*>(d_texc), reinterpret_cast<const float4*>(d_texd),
            reinterpret_cast<const float4*>(d_rast), reinterpret_cast<const float4*>(d_rast_db), uv_nograd_faces, V, F, H, W, d_pos, d_vnormal, vhap_g_debug_flags);
        VHAP_LAUNCH_CHECK();
        return VHAP_OK;
    }
    gbuffer_bwd_kernel<<<vhap_cdiv((long long)B * F * GB_LANES, 256), 256, 0, vhap_stream(stream)>>>(
        reinterpret_cast<const float4*>(pos), tri, vnormal, reinterpret_cast<const float2*>(uv), tri_uv,
        reinterpret_cast<const float4*>(rast), d_normal, reinterpret_cast<const float2*>(d_texc), reinterpret_cast<const float4*>(d_texd),
        reinterpret_cast<const float4*>(d_rast), reinterpret_cast<const float4*>(d_rast_db), uv_nograd_faces, B, V, F, H, W, d_pos, d_vnormal);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
