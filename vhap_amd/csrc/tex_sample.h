// Trilinear mip-mapped texture sampling: the per-sample arithmetic shared by the stand-alone texture op (texture.hip) and the
// deferred-shading kernels (raster.hip: forward inside the rasteriser; deferred.hip: backward), so that all of them compute the
// same bits.  Semantics: header of texture.hip (dr.texture, 'linear-mipmap-linear', boundary 'wrap', render_nvdiffrast.py:399).
#pragma once
#include <cstdlib>
#include "common.h"

namespace {

constexpr int MAX_LEVELS = 14;

struct TexDesc {
    int TB, H, W, C, L;              // L = number of levels above level 0
    long long off[MAX_LEVELS + 1];   // float offset of level l (l >= 1) inside the mip buffer, per texture copy
    long long per_tex;               // floats per texture copy in the mip buffer
};

int num_levels(int H, int W) {
    int L = 0;
    while (H > 1 && W > 1 && H % 2 == 0 && W % 2 == 0 && L < MAX_LEVELS) { H >>= 1; W >>= 1; L++; }
    return L;
}

TexDesc make_desc(int TB, int H, int W, int C) {
    TexDesc d;
    d.TB = TB; d.H = H; d.W = W; d.C = C; d.L = num_levels(H, W);
    long long o = 0;
    d.off[0] = 0;
    for (int l = 1; l <= d.L; l++) {
        d.off[l] = o;
        // (device code computes this offset in closed form: level_off)
        if (o != (long long)(((unsigned)(H * W) - ((unsigned)(H * W) >> (2 * (l - 1)))) / 3u) * C) abort();
        o += (long long)(H >> l) * (W >> l) * C;
    }
    d.per_tex = o;
    return d;
}

struct Taps {
    int i00, i10, i01, i11;  // texel indices (already multiplied by C)
    float fx, fy;
    int x0, y0, x1, y1;      // texel coordinates after wrapping
};

__device__ __forceinline__ Taps make_taps(float u, float v, int w, int h, int C) {
    Taps t;
    u = u - floorf(u);
    v = v - floorf(v);
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float x0f = floorf(x), y0f = floorf(y);
    t.fx = x - x0f;
    t.fy = y - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;
    int x1 = x0 + 1, y1 = y0 + 1;
    if (x0 < 0) x0 += w;
    if (y0 < 0) y0 += h;
    if (x1 >= w) x1 -= w;
    if (y1 >= h) y1 -= h;
    // guard against u == 1.0 after rounding (x0 == w)
    if (x0 >= w) x0 -= w;
    if (y0 >= h) y0 -= h;
    t.x0 = x0; t.y0 = y0; t.x1 = x1; t.y1 = y1;
    t.i00 = (y0 * w + x0) * C; t.i10 = (y0 * w + x1) * C;
    t.i01 = (y1 * w + x0) * C; t.i11 = (y1 * w + x1) * C;
    return t;
}

struct LevelSel {
    int l0;        // lower level
    float f;       // blend factor toward l0+1 (0 when only one level is sampled)
    bool two;      // sample level l0+1 as well
    bool diff;     // level strictly inside (0, L): gradient flows to uv_da
    float lambda, l2n_sqrt, A, B, Cq, sx, sy, tx, ty;
};

__device__ __forceinline__ LevelSel select_level(const float4 da, int Wt, int Ht, int L) {
    LevelSel s;
    s.sx = da.x * (float)Wt; s.sy = da.y * (float)Wt;
    s.tx = da.z * (float)Ht; s.ty = da.w * (float)Ht;
    s.A = s.sx * s.sx + s.tx * s.tx;
    s.B = s.sy * s.sy + s.ty * s.ty;
    s.Cq = s.sx * s.sy + s.tx * s.ty;
    const float l2b = 0.5f * (s.A + s.B);
    const float l2n = 0.25f * (s.A - s.B) * (s.A - s.B) + s.Cq * s.Cq;
    s.l2n_sqrt = sqrtf(l2n);
    s.lambda = l2b + s.l2n_sqrt;
    float level = 0.5f * log2f(fmaxf(s.lambda, 1e-30f));
    s.diff = level > 0.0f && level < (float)L;
    level = fminf(fmaxf(level, 0.0f), (float)L);
    int l0 = (int)floorf(level);
    if (l0 > L - 1) l0 = L - 1;
    if (l0 < 0) l0 = 0;
    s.l0 = l0;
    s.f = level - (float)l0;
    s.two = L > 0;
    if (L == 0) s.f = 0.0f;
    return s;
}

// off[l] for a level chosen per lane, by arithmetic: indexing the table with a lane-varying l makes the compiler fetch it from the
// kernel-argument segment with a global load -- one more dependent round trip in front of each level's taps (two per sample; found in the
// ISA of the pixel kernels, profiles/r04_call22_step_sq_pmc.json: they sit at s_waitcnt 60-80 % of their wave-cycles).  H and W are
// divisible by 2^L (num_levels), so (H >> k)(W >> k) = HW >> 2k exactly and off[l] = C * sum_{k=1..l-1} HW / 4^k = C * (HW - HW / 4^(l-1)) / 3.
__device__ __forceinline__ size_t level_off(const TexDesc& D, int l) {
    const unsigned M = (unsigned)D.H * (unsigned)D.W;
    return (size_t)((M - (M >> (2 * (l - 1)))) / 3u) * (unsigned)D.C;
}
__device__ __forceinline__ const float* level_ptr(const float* tex, const float* mips, const TexDesc& D, int tb, int l) {
    return l == 0 ? tex + (size_t)tb * D.H * D.W * D.C : mips + (size_t)tb * D.per_tex + level_off(D, l);
}
__device__ __forceinline__ float* level_ptr_w(float* tex, float* mips, const TexDesc& D, int tb, int l) {
    return l == 0 ? tex + (size_t)tb * D.H * D.W * D.C : mips + (size_t)tb * D.per_tex + level_off(D, l);
}

template <int C>
__device__ __forceinline__ void bilinear_bwd(const float* __restrict__ T, float* __restrict__ G, const Taps& t,
                                             const float (&g)[C], float wgt, float& gfx, float& gfy, float (&val)[C]) {
    gfx = 0.f;
    gfy = 0.f;
    const float w00 = (1.f - t.fx) * (1.f - t.fy), w10 = t.fx * (1.f - t.fy), w01 = (1.f - t.fx) * t.fy, w11 = t.fx * t.fy;
#pragma unroll
    for (int k = 0; k < C; k++) {
        const float a00 = T[t.i00 + k], a10 = T[t.i10 + k], a01 = T[t.i01 + k], a11 = T[t.i11 + k];
        const float top = a00 + t.fx * (a10 - a00), bot = a01 + t.fx * (a11 - a01);
        val[k] = top + t.fy * (bot - top);
        const float gk = g[k] * wgt;
        gfx += gk * ((1.f - t.fy) * (a10 - a00) + t.fy * (a11 - a01));
        gfy += gk * (bot - top);
        if (G && gk != 0.f) {
            atomicAdd(&G[t.i00 + k], w00 * gk);
            atomicAdd(&G[t.i10 + k], w10 * gk);
            atomicAdd(&G[t.i01 + k], w01 * gk);
            atomicAdd(&G[t.i11 + k], w11 * gk);
        }
    }
}


// ---- uv-space binning of the texture gradient (texture.hip: vhap_texture_grad_binned): shared with deferred.hip, which fills the tile
// histogram while it computes d_albedo ----
constexpr int TG_MAX_NT = 64;       // at most 64 x 64 tiles: the per-workgroup histograms are 2 x 16 KiB of LDS
inline int texbin_nt(int Ht, int Wt) {
    int nt = 8;
    const int m = Wt > Ht ? Wt : Ht;
    while (nt < TG_MAX_NT && m / nt > 32) nt *= 2;
    return nt;
}
__device__ __forceinline__ int tile_of(float2 c, int NT) {
    const float uf = c.x - floorf(c.x), vf = c.y - floorf(c.y);
    const int tx = min(NT - 1, (int)(uf * (float)NT)), ty = min(NT - 1, (int)(vf * (float)NT));
    return ty * NT + tx;
}
struct TexBinWs {
    size_t counts, tilemax, cursors, offsets, list, total;
};
TexBinWs texbin_layout(long long npix) {
    TexBinWs l;
    const size_t n = (size_t)TG_MAX_NT * TG_MAX_NT;
    l.counts = 0;
    l.tilemax = l.counts + n * 4;
    l.cursors = l.tilemax + n * 4;
    l.offsets = l.cursors + n * 4;
    l.list = l.offsets + (n + 64) * 4;
    l.total = l.list + (size_t)npix * 4;
    return l;
}


// ---- fetch / finish split of one trilinear sample, for the pixel kernels (raster.hip mode 2, deferred.hip).  A covered wave's life is
// a chain of dependent gathers, so ALL eight taps are issued in one batch (the coarser level's too: l0 + 1 <= L is a valid level
// whatever the blend factor; it was fetched behind a per-lane branch, a round trip of its own) and the caller may issue further
// independent loads before it consumes them.  tex_value / tex_bwd_uv evaluate the same expressions as tex_sample / bilinear_bwd.
template <int C>
struct TexFetch {
    LevelSel s;
    Taps t0, t1;
    int w0, h0, w1, h1;
    float a0[4][C], a1[4][C];
};
template <int C>
__device__ __forceinline__ TexFetch<C> tex_fetch(const float* __restrict__ tex, const float* __restrict__ mips, const TexDesc& D, int tb,
                                                 const float2 c, const float4 da) {
    TexFetch<C> f;
    f.s = select_level(da, D.W, D.H, D.L);
    f.w0 = D.W >> f.s.l0; f.h0 = D.H >> f.s.l0;
    f.t0 = make_taps(c.x, c.y, f.w0, f.h0, C);
    const float* T0 = level_ptr(tex, mips, D, tb, f.s.l0);
#pragma unroll
    for (int k = 0; k < C; k++) {
        f.a0[0][k] = T0[f.t0.i00 + k]; f.a0[1][k] = T0[f.t0.i10 + k]; f.a0[2][k] = T0[f.t0.i01 + k]; f.a0[3][k] = T0[f.t0.i11 + k];
    }
    // (no branch: a join would make the compiler wait for the taps at the end of the block.  Without mip levels -- L = 0, uniform --
    // the "coarser" level is level 0 again: valid addresses, values never used)
    const int l1 = f.s.two ? f.s.l0 + 1 : f.s.l0;
    f.w1 = D.W >> l1; f.h1 = D.H >> l1;
    f.t1 = make_taps(c.x, c.y, f.w1, f.h1, C);
    const float* T1 = level_ptr(tex, mips, D, tb, l1);
#pragma unroll
    for (int k = 0; k < C; k++) {
        f.a1[0][k] = T1[f.t1.i00 + k]; f.a1[1][k] = T1[f.t1.i10 + k]; f.a1[2][k] = T1[f.t1.i01 + k]; f.a1[3][k] = T1[f.t1.i11 + k];
    }
    return f;
}
template <int C>
__device__ __forceinline__ void tex_value(const TexFetch<C>& f, float (&res)[C]) {
    const bool two = f.s.two && f.s.f > 0.0f;
#pragma unroll
    for (int k = 0; k < C; k++) {
        const float top = f.a0[0][k] + f.t0.fx * (f.a0[1][k] - f.a0[0][k]);
        const float bot = f.a0[2][k] + f.t0.fx * (f.a0[3][k] - f.a0[2][k]);
        const float r0 = top + f.t0.fy * (bot - top);
        const float top1 = f.a1[0][k] + f.t1.fx * (f.a1[1][k] - f.a1[0][k]);
        const float bot1 = f.a1[2][k] + f.t1.fx * (f.a1[3][k] - f.a1[2][k]);
        const float c1 = top1 + f.t1.fy * (bot1 - top1);
        res[k] = two ? (1.0f - f.s.f) * r0 + f.s.f * c1 : r0;
    }
}
template <int C>
__device__ __forceinline__ void bilinear_fetched(const float (&a)[4][C], const Taps& t, const float (&g)[C], float wgt, float& gfx, float& gfy,
                                                 float (&val)[C]) {
    gfx = 0.f;
    gfy = 0.f;
#pragma unroll
    for (int k = 0; k < C; k++) {
        const float a00 = a[0][k], a10 = a[1][k], a01 = a[2][k], a11 = a[3][k];
        const float top = a00 + t.fx * (a10 - a00), bot = a01 + t.fx * (a11 - a01);
        val[k] = top + t.fy * (bot - top);
        const float gk = g[k] * wgt;
        gfx += gk * ((1.f - t.fy) * (a10 - a00) + t.fy * (a11 - a01));
        gfy += gk * (bot - top);
    }
}
// gradient w.r.t. uv and uv_da of the fetched sample for the upstream gradient g[C] (tex_sample_bwd_uv without gradient images)
template <int C>
__device__ __forceinline__ void tex_bwd_uv(const TexFetch<C>& f, const TexDesc& D, const float (&g)[C], float2& guv, float4& gda,
                                           float (&val)[C]) {
    const LevelSel& s = f.s;
    const bool two = s.two && s.f > 0.0f;
    float gfx0, gfy0, c0[C], gfx1, gfy1, c1[C];
    bilinear_fetched<C>(f.a0, f.t0, g, two ? 1.0f - s.f : 1.0f, gfx0, gfy0, c0);
    bilinear_fetched<C>(f.a1, f.t1, g, s.f, gfx1, gfy1, c1);
    guv.x = gfx0 * (float)f.w0;
    guv.y = gfy0 * (float)f.h0;
    gda = make_float4(0.f, 0.f, 0.f, 0.f);
    if (two) {
        guv.x += gfx1 * (float)f.w1;
        guv.y += gfy1 * (float)f.h1;
    }
#pragma unroll
    for (int k = 0; k < C; k++) val[k] = two ? (1.0f - s.f) * c0[k] + s.f * c1[k] : c0[k];
    if (two && s.diff) {
        float gf = 0.f;
#pragma unroll
        for (int k = 0; k < C; k++) gf += g[k] * (c1[k] - c0[k]);
        const float glam = gf * 0.5f / (s.lambda * 0.69314718056f);
        const float q = s.l2n_sqrt > 0.f ? 0.5f / s.l2n_sqrt : 0.f;
        const float gl2n = glam * q;
        const float gA = 0.5f * glam + gl2n * 0.5f * (s.A - s.B);
        const float gB = 0.5f * glam - gl2n * 0.5f * (s.A - s.B);
        const float gC = gl2n * 2.0f * s.Cq;
        const float gsx = 2.f * s.sx * gA + s.sy * gC, gsy = 2.f * s.sy * gB + s.sx * gC;
        const float gtx = 2.f * s.tx * gA + s.ty * gC, gty = 2.f * s.ty * gB + s.tx * gC;
        gda = make_float4(gsx * (float)D.W, gsy * (float)D.W, gtx * (float)D.H, gty * (float)D.H);
    }
}

// value of ONE trilinear sample (the body of texture_fwd_kernel): res[C]
template <int C>
__device__ __forceinline__ void tex_sample(const float* __restrict__ tex, const float* __restrict__ mips, const TexDesc& D, int tb,
                                           const float2 c, const float4 da, float (&res)[C]) {
    const LevelSel s = select_level(da, D.W, D.H, D.L);
    const Taps t0 = make_taps(c.x, c.y, D.W >> s.l0, D.H >> s.l0, C);
    const float* T0 = level_ptr(tex, mips, D, tb, s.l0);
#pragma unroll
    for (int k = 0; k < C; k++) {
        const float top = T0[t0.i00 + k] + t0.fx * (T0[t0.i10 + k] - T0[t0.i00 + k]);
        const float bot = T0[t0.i01 + k] + t0.fx * (T0[t0.i11 + k] - T0[t0.i01 + k]);
        res[k] = top + t0.fy * (bot - top);
    }
    if (s.two && s.f > 0.0f) {
        const Taps t1 = make_taps(c.x, c.y, D.W >> (s.l0 + 1), D.H >> (s.l0 + 1), C);
        const float* T1 = level_ptr(tex, mips, D, tb, s.l0 + 1);
#pragma unroll
        for (int k = 0; k < C; k++) {
            const float top = T1[t1.i00 + k] + t1.fx * (T1[t1.i10 + k] - T1[t1.i00 + k]);
            const float bot = T1[t1.i01 + k] + t1.fx * (T1[t1.i11 + k] - T1[t1.i01 + k]);
            const float c1 = top + t1.fy * (bot - top);
            res[k] = (1.0f - s.f) * res[k] + s.f * c1;
        }
    }
}

// gradient of ONE trilinear sample w.r.t. uv and uv_da for the upstream gradient g[C] (the body of texture_bwd_kernel with
// mip-mapping); G0 / G1: optional gradient images of the two levels (atomics), normally null here
template <int C>
__device__ __forceinline__ void tex_sample_bwd_uv(const float* __restrict__ tex, const float* __restrict__ mips, const TexDesc& D, int tb,
                                                  const float2 c, const float4 da, const float (&g)[C], float* __restrict__ d_tex,
                                                  float* __restrict__ d_mips, float2& guv, float4& gda, bool want_da,
                                                  float (&val)[C]) {
    guv = make_float2(0.f, 0.f);
    gda = make_float4(0.f, 0.f, 0.f, 0.f);
    const LevelSel s = select_level(da, D.W, D.H, D.L);
    const int w0 = D.W >> s.l0, h0 = D.H >> s.l0;
    const Taps t0 = make_taps(c.x, c.y, w0, h0, C);
    const bool two = s.two && s.f > 0.0f;
    float gfx0, gfy0, c0[C];
    bilinear_bwd<C>(level_ptr(tex, mips, D, tb, s.l0), d_tex ? level_ptr_w(d_tex, d_mips, D, tb, s.l0) : nullptr, t0, g,
                    two ? 1.0f - s.f : 1.0f, gfx0, gfy0, c0);
    guv.x = gfx0 * (float)w0;
    guv.y = gfy0 * (float)h0;
#pragma unroll
    for (int k = 0; k < C; k++) val[k] = c0[k];            // the sample's value, same expressions as tex_sample()
    if (two) {
        const int w1 = D.W >> (s.l0 + 1), h1 = D.H >> (s.l0 + 1);
        const Taps t1 = make_taps(c.x, c.y, w1, h1, C);
        float gfx1, gfy1, c1[C];
        bilinear_bwd<C>(level_ptr(tex, mips, D, tb, s.l0 + 1), d_tex ? level_ptr_w(d_tex, d_mips, D, tb, s.l0 + 1) : nullptr, t1, g, s.f,
                        gfx1, gfy1, c1);
        guv.x += gfx1 * (float)w1;
        guv.y += gfy1 * (float)h1;
#pragma unroll
        for (int k = 0; k < C; k++) val[k] = (1.0f - s.f) * c0[k] + s.f * c1[k];
        if (s.diff && want_da) {
            float gf = 0.f;
#pragma unroll
            for (int k = 0; k < C; k++) gf += g[k] * (c1[k] - c0[k]);
            // level = 0.5*log2(lambda); lambda = (A+B)/2 + sqrt((A-B)^2/4 + C^2)
            const float glam = gf * 0.5f / (s.lambda * 0.69314718056f);
            const float q = s.l2n_sqrt > 0.f ? 0.5f / s.l2n_sqrt : 0.f;  // d sqrt(l2n) / d l2n
            const float gl2n = glam * q;
            const float gA = 0.5f * glam + gl2n * 0.5f * (s.A - s.B);
            const float gB = 0.5f * glam - gl2n * 0.5f * (s.A - s.B);
            const float gC = gl2n * 2.0f * s.Cq;
            const float gsx = 2.f * s.sx * gA + s.sy * gC, gsy = 2.f * s.sy * gB + s.sx * gC;
            const float gtx = 2.f * s.tx * gA + s.ty * gC, gty = 2.f * s.ty * gB + s.tx * gC;
            gda = make_float4(gsx * (float)D.W, gsy * (float)D.W, gtx * (float)D.H, gty * (float)D.H);
        }
    }
}

}  // namespace
